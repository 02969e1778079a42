#pragma once
#include "common.h"
// Factor the 128x128 diagonal block A (lower Cholesky, in place) and, when inv != NULL, write its
// inverse (lower, upper part zero) to inv.  *info (device) receives col0 + j + 1 on a bad pivot.
// ablate != 0 skips phases (timing diagnostics only: bit 0 factor loop, 1 inverse, 3 diagonal step).
// nexp > 1: the same leaf of nexp batched problems in one launch (A + e * eA, inv + e * eInv, info[e]).
template <typename T> int pg_leaf(hipStream_t st, T* A, long lda, T* inv, long ldi, int* info, int col0, int ablate = 0, int nexp = 1,
                                  long eA = 0, long eInv = 0);
// The same leaf as one half of the flag-coupled chain (chainstep.hip): waits for *ready >= want, factors, sets *done.
struct CsWait;
struct CsBatch;
// *early (nullable): raised by the third form once the first four block rows (64 rows) of the tile's inverse are in memory.
bool pg_leaf_has_early();
template <typename T> int pg_leaf_sync(hipStream_t st, T* A, long lda, T* inv, int* info, int col0, int* ready, int want, int* done,
                                       const CsWait& tmo, int* early = nullptr, const CsBatch* cb = nullptr);   // cb: one workgroup per expert
