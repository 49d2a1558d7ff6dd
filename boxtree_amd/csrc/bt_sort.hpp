// LSD radix sort of (key, u32 value) pairs, 8-bit digits, one sweep per digit.
#pragma once

#include "bt_common.hpp"

namespace bt {

// Sorts pairs on bits [begin_bit, end_bit).  (ka, va) hold the input; (kb, vb)
// are equally sized scratch.  On return *in_b tells which pair of buffers holds
// the sorted result.  identity_vals: the values are 0..n-1; the first pass
// synthesises them instead of reading va (va is still used as scratch).
template <class KeyT>
int radix_sort_pairs(bt_context *ctx, KeyT *ka, uint32_t *va, KeyT *kb, uint32_t *vb,
                     int64_t n, int begin_bit, int end_bit, bool identity_vals, bool *in_b);

extern template int radix_sort_pairs<uint64_t>(bt_context *, uint64_t *, uint32_t *, uint64_t *,
                                               uint32_t *, int64_t, int, int, bool, bool *);
extern template int radix_sort_pairs<uint32_t>(bt_context *, uint32_t *, uint32_t *, uint32_t *,
                                               uint32_t *, int64_t, int, int, bool, bool *);

// Keys only: sorts 64-bit words on bits [begin_bit, end_bit) (the bits below ride along,
// e.g. a user id packed under a Morton path); kb is scratch, *in_b tells where the result
// is.  8- or 9-bit digits, whichever needs fewer passes (radix_sort_keys_plan).
int radix_sort_keys(bt_context *ctx, uint64_t *ka, uint64_t *kb, int64_t n, int begin_bit,
                    int end_bit, bool *in_b);
int radix_sort_keys_plan(int nbits, int *digit_bits);       // -> number of passes

}  // namespace bt
