// plan.hpp — the opaque splahip_plan: work decomposition of one device CSR matrix (built by mxv.hip, used by mxv / vxm).
#pragma once

#include <stdint.h>

struct splahip_plan_s {
    uint32_t  n_rows  = 0;
    uint32_t  nnz     = 0;
    uint32_t  n_items = 0;
    uint32_t  n_long  = 0;         // rows cut into segment items
    uint32_t* d_item_row   = nullptr;// n_items + 1
    uint32_t* d_item_k0    = nullptr;// n_items + 1
    uint32_t* d_partial    = nullptr;// n_items words of scratch (per-segment partial results)
    uint32_t* d_partial_ok = nullptr;// n_items words of scratch (vxm: does the segment hold a product)
};
