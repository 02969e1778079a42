#pragma once
#include "common.h"
#define PG_CS_NCRIT 8   // workgroups that own block row k+1 (and so publish tile (k+1, k+1)) in a rows kernel
// Flag-coupled chain step (chainstep.hip): the rows below tile (k, k) of the outer panel that starts at column o0 -- the next block
// column's update by the panel's earlier columns, then (after *done_k) the solve against inv_k, then the last 128 columns of the
// update.  has_next = 0: the panel's last step (solve only).  n, k0, o0 multiples of 128.
template <typename T>
int pg_rowstep(hipStream_t st, T* A, long lda, int n, int o0, int k0, int has_next, const T* inv, int* done_k, int* brow_k,
               int* diag_next, int* tmo, int* info);
int pg_flagset(hipStream_t st, int* flag, int value);
