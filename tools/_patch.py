"""Tiny helper for scripted source edits: replace the text between two anchors, failing loudly when an anchor is missing,
ambiguous or out of order (development tooling, not product code)."""


def replace_between(s, start, end, new, include_end=False):
    assert s.count(start) == 1, "start anchor occurs %d times: %r" % (s.count(start), start[:60])
    i = s.index(start)
    j = s.index(end, i + len(start))
    if include_end:
        j += len(end)
    return s[:i] + new + s[j:]


def replace_once(s, old, new):
    assert s.count(old) == 1, "anchor occurs %d times: %r" % (s.count(old), old[:60])
    return s.replace(old, new)
