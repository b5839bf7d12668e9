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
    // call workspace, grown on demand and kept with the plan so steady-state calls do not allocate (vxm pull form)
    void*     d_ws         = nullptr;
    size_t    ws_bytes     = 0;
    // hot-column cache (splahip_csr_plan_attach_columns): the matrix's column ids re-encoded for spmv_hot_kernel
    uint32_t  n_hot        = 0;      // columns whose values are kept in LDS
    uint32_t* d_hot_cols   = nullptr;// n_hot column ids, in LDS-slot order
    uint32_t* d_hot_values = nullptr;// n_hot words of scratch, gathered per launch
    uint32_t* d_Aj_enc     = nullptr;// nnz words: bit 31 set -> low bits are the LDS slot, else the column id
};
