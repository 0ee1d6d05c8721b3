// klr_kmers.h -- library-internal view of the k-mer class tables (klr_kmers.cpp); the public part is in include/kmerlr.h
#pragma once
#include <cstdint>
#include <cstdio>

#include "../../include/kmerlr.h"

extern "C" {
// code -> 0-based class id, one table per length M..N (map_off[N - M + 2] offsets), base = letters per position
int klr_kmer_classes_tables(const klr_kmer_classes* kc, const uint32_t** map, const uint64_t** map_off, int* base);
const klr_kmer_spec* klr_kmer_classes_spec(const klr_kmer_classes* kc);
int klr_kmer_classes_max_ambiguous(const klr_kmer_classes* kc, int k);
}
