"""CPU: the k-mer class tables (klr_kmers.cpp, host code of the product library) against the only numbering the
reference holds: TestKmers1 (kmerLr_test.go:30-66) addresses six classes of its M = 4, N = 8, revcomp,
gapped-nucleotide counter by their INDEX in the list of classes observed in kmerLr_test.fa -- 4671, 4672, 5068, 5486,
19270, 57071.  The observed classes in this library's enumeration order put the same six names at the same six places:
the order of the classes (by length, then by gonetics' class id) is pinned at those points, not only their names and
counts."""
import os

import kmerlr_amd as K
from tests.conftest import GOLDEN


def test_kmers1_class_indices_among_the_observed_classes():
    seqs, cur = [], []
    for line in open(os.path.join(GOLDEN, "kmerLr_test.fa")):
        line = line.strip()
        if line.startswith(">"):
            if cur:
                seqs.append("".join(cur)); cur = []
        elif line:
            cur.append(line)
    seqs.append("".join(cur))
    kc = K.KmerClasses(4, 8, revcomp=True, alphabet="gapped-nucleotide")
    observed = set()
    for s in seqs:
        s = s.lower()
        for k in range(4, 9):
            for p in range(len(s) - k + 1):
                w = s[p:p + k]
                if any(ch not in "acgt" for ch in w):
                    continue
                for mask in range(1 << (k - 2)):                    # every choice of gaps between the two end letters
                    pat = w[0] + "".join("n" if (mask >> i) & 1 else w[1 + i] for i in range(k - 2)) + w[-1]
                    c = kc.find(pat)
                    assert c >= 0, pat
                    observed.add(c)
    order = sorted(observed)                                        # compile_training_data lists them in class order
    place = {c: i for i, c in enumerate(order)}
    for name, index in (("gntanc|gntanc", 4671), ("gntcaa|ttganc", 4672), ("aaagaaa|tttcttt", 5068), ("aagannt|anntctt", 5486),
                        ("aacgcgna|tncgcgtt", 19270), ("tgaatgca|tgcattca", 57071)):      # kmerLr_test.go:41-44
        c = kc.find(name)
        assert c >= 0 and kc.name(c) == name
        assert place[c] == index, (name, place[c], index)
    kc.close()
