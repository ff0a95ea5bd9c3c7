#!/usr/bin/env python3
"""Re-encode the reference's alignment pickles as plain FASTA under phylo_amd/data/.

The reference keeps its alignments as pickled dict[str, str] (taxon -> sequence), loaded by
runner.py:117-171 via pandas.read_pickle.  Pickles do not belong in a repo that ships to a GPU
box, so this script (run once, in the build container, where /root/reference exists) writes the
same taxa in the same order as FASTA text.  Only data is converted; no reference code is used.
"""
import io
import os
import pickle
import sys
import zipfile

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "phylo_amd", "data")


def write_fasta(path, d):
    with open(path, "w") as f:
        for name, seq in d.items():
            f.write(">%s\n" % name)
            for i in range(0, len(seq), 80):
                f.write(seq[i:i + 80] + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    for src, dst in (("primate.p", "primate.fa"), ("primates_small.p", "primates_small.fa"),
                     ("fish.p", "fish.fa")):
        with open(os.path.join(REF, src), "rb") as f:
            write_fasta(os.path.join(OUT, dst), pickle.load(f))
    z = zipfile.ZipFile(os.path.join(REF, "hohna_dataset_pickle.zip"))
    for n in range(1, 9):
        d = pickle.load(io.BytesIO(z.read("DS%d.pickle" % n)))
        write_fasta(os.path.join(OUT, "hohna_DS%d.fa" % n), d)
    return 0


if __name__ == "__main__":
    sys.exit(main())
