"""Dataset loading and one-hot / gap encoding (reference: runner.py:83-192).

The reference selects a dataset by `exec(args.dataset + ' = True')` (runner.py:81) and reads pickled
dict[str, str] alignments with pandas; here the same dataset names map to FASTA files committed under
phylo_amd/data/ (re-encoded by tools/convert_datasets.py), and unknown names raise ValueError.
"""
from __future__ import annotations

import os
import random

import numpy as np

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# runner.py:83-96
Alphabet_dir = {'A': [1, 0, 0, 0], 'C': [0, 1, 0, 0], 'G': [0, 0, 1, 0], 'T': [0, 0, 0, 1]}
alphabet_dir = {'a': [1, 0, 0, 0], 'c': [0, 1, 0, 0], 'g': [0, 0, 1, 0], 't': [0, 0, 0, 1]}
Alphabet_dir_blank = {'A': [1, 0, 0, 0], 'C': [0, 1, 0, 0], 'G': [0, 0, 1, 0], 'T': [0, 0, 0, 1],
                      '-': [1, 1, 1, 1], '?': [1, 1, 1, 1]}
alphabet = np.array([[1., 0., 0., 0.], [0., 1., 0., 0.], [0., 0., 1., 0.], [0., 0., 0., 1.]])
# NOT in the reference (opt-in, `load_dataset(name, ambiguity='iupac')` / `runner.py --ambiguity iupac`): the IUPAC
# nucleotide codes as indicator rows over (A, C, G, T), so that DS7 (28 'N' cells) loads.  'N' gets the
# same all-ones row as '-' and '?', the others the set they stand for.
_IUPAC = {'N': 'ACGT', 'X': 'ACGT', 'R': 'AG', 'Y': 'CT', 'S': 'CG', 'W': 'AT', 'K': 'GT', 'M': 'AC',
          'B': 'CGT', 'D': 'AGT', 'H': 'ACT', 'V': 'ACG', 'U': 'T'}
Alphabet_dir_iupac = dict(Alphabet_dir_blank)
for _c, _set in _IUPAC.items():
    Alphabet_dir_iupac[_c] = [1 if b in _set else 0 for b in 'ACGT']


def load_fasta(path):
    """Returns (taxa names, sequences) in file order."""
    names, seqs = [], []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            if line.startswith('>'):
                names.append(line[1:])
                seqs.append([])
            else:
                seqs[-1].append(line)
    return names, [''.join(s) for s in seqs]


def form_dataset_from_strings(genome_strings, alphabet_dir, alphabet_num=4):
    """runner.py:107-115.  Same return structure ({'taxa': ['S0', ...], 'genome': [N,S,4] f64}) and the
    same error behaviour: a character missing from `alphabet_dir` raises KeyError (e.g. 'N' in DS7)."""
    table = np.zeros((256, alphabet_num))
    known = np.zeros(256, dtype=bool)
    for ch, row in alphabet_dir.items():
        table[ord(ch)] = row
        known[ord(ch)] = True
    n, s = len(genome_strings), len(genome_strings[0])
    codes = np.empty((n, s), dtype=np.uint8)
    for i, g in enumerate(genome_strings):
        row = np.frombuffer(g.encode('latin-1'), dtype=np.uint8)
        if row.shape[0] != s:
            raise ValueError("sequence %d has length %d, expected %d" % (i, row.shape[0], s))
        bad = ~known[row]
        if bad.any():
            raise KeyError(chr(int(row[np.argmax(bad)])))
        codes[i] = row
    genomes_NxSxA = table[codes]
    taxa = ['S' + str(i) for i in range(n)]
    return {'taxa': taxa, 'genome': genomes_NxSxA}


def simulateDNA(nsamples, seqlength, alphabet):
    """runner.py:100-104 (iid uniform one-hot rows, python's global RNG)."""
    genomes_NxSxA = np.zeros([nsamples, seqlength, alphabet.shape[0]])
    for n in range(nsamples):
        genomes_NxSxA[n] = np.array([random.choice(alphabet) for i in range(seqlength)])
    return genomes_NxSxA


def synthetic_alignment(n_taxa, n_sites, seed=20260005):
    """numpy default_rng(seed).integers(0,4,(N,S)) -> one-hot fp64, no gaps."""
    codes = np.random.default_rng(seed).integers(0, 4, size=(n_taxa, n_sites))
    return {'taxa': ['S' + str(i) for i in range(n_taxa)], 'genome': alphabet[codes]}


_FASTA = {
    'primate_data': ('primate.fa', Alphabet_dir_blank),            # runner.py:163-166
    'primate_data_wang': ('primates_small.fa', Alphabet_dir),      # runner.py:168-171
    'hohna_data': ('hohna_DS1.fa', Alphabet_dir_blank),            # runner.py:117-120
}
for _i in range(1, 9):
    _FASTA['hohna_data_%d' % _i] = ('hohna_DS%d.fa' % _i, Alphabet_dir_blank)


def load_dataset(name, ambiguity='error'):
    """Dataset table replacing runner.py:81/117-192.  Returns the reference's datadict.
    ambiguity='error' (default) is the reference's behaviour: a character outside the dataset's alphabet raises KeyError
    (hohna_data_7 holds 'N', runner.py:91-96,107-115); ambiguity='iupac' encodes IUPAC ambiguity codes as indicator rows."""
    if ambiguity not in ('error', 'iupac'):
        raise ValueError("ambiguity must be 'error' or 'iupac', got %r" % (ambiguity,))
    if name in _FASTA:
        fname, adir = _FASTA[name]
        if ambiguity == 'iupac':
            adir = dict(Alphabet_dir_iupac, **adir)
        _, seqs = load_fasta(os.path.join(DATA_DIR, fname))
        return form_dataset_from_strings(seqs, adir)
    if name == 'load_strings':                                     # runner.py:182-184
        genome_strings = ['ACTTTGAGAG', 'ACTTTGACAG', 'ACTTTGACTG', 'ACTTTGACTC']
        return form_dataset_from_strings(genome_strings, Alphabet_dir)
    if name == 'simulate_data':                                    # runner.py:174-179
        data_NxSxA = simulateDNA(3, 5, alphabet)
        return {'taxa': ['S' + str(i) for i in range(data_NxSxA.shape[0])], 'genome': data_NxSxA}
    if name == 'corona_data':                                      # runner.py:159-160; .MISSING_LARGE_BLOBS
        raise FileNotFoundError("data/coronavirus.p is not distributed with the reference")
    raise ValueError("unknown dataset %r" % (name,))
