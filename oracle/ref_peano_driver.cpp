/* C entry points over the reference's own Peano-Hilbert key (libgadget/utils/peano.cpp, compiled where it lies by
 * `make -C oracle ref` into oracle/_ref/libpeano_ref.so).  Test infrastructure: tests/test_peano_ref_cpu.py uses it to check
 * that the top tree handed to shq_tree_build_domain as geometry is the tree the reference's key arithmetic describes. */
#include "utils/peano.h"

extern "C" unsigned long long ref_peano_hilbert_key(int x, int y, int z, int bits) { return peano_hilbert_key(x, y, z, bits); }
extern "C" unsigned long long ref_PEANO(const double *pos, double BoxSize) { return PEANO(pos, BoxSize); }
