// cob.hip — the "column-ordered blocks" copy of a device CSR matrix: built once per matrix on the device, kept with the plan
// (part of the AccCsr decoration, like the wave-tile items and the sliced-ELL copy), and the masked SpMV / pull-form vxm over it.
//
// Semantics served: reference Algo_mxv_masked_cpu, src/cpu/cpu_mxv.hpp:88-103, and the dense-frontier case of Algo_vxm_masked_cpu,
// src/cpu/cpu_vxm.hpp:92-125, for an op_add with an LDS atomic (mxv_cob.hpp: cob_acc_kind).  Kernels: mxv_cob.hpp.
//
// Build (all on the device; the host reads a few counters and one byte per 8 rows):
//   0. rows -> ranges     row degrees; the rows of degree >= max(16, 2 x average) sorted by degree become HUB ranges of R rows (at most
//                         SPLAHIP_COB_HUB_RANGES = 16 ranges); the other rows that hold an entry fill NATURAL ranges of <= R such rows in row
//                         order (boundaries at multiples of 8 rows, chosen by one greedy pass on the host over per-8-row counts)
//   1. cob_key_kernel     key(k) = ((range << cbits | column) << 15) | slot_in_range for every stored entry, by one binary search of
//                         the entry's 2048-entry tile in Ap and one inside the tile's row span
//   2. a stable LSD radix sort of the keys on the (range, column) bits only, carrying the value word (rocprim, one-off per matrix):
//      inside a (range, column) run the entries keep their order
//   3. cob_group_kernel   first sorted entry of every (range, column window) group by binary search; chunks per group; scan
//   3b. cob_weight_kernel slots per entry: itself + one bridging slot per 255 columns of a gap of more than 255 to the entry before it (the column stream holds
//                         8-bit deltas); their running sum places every entry
//   4. cob_scatter_kernel one workgroup per chunk: pack sl / dl / ax / hdr (16-bit slot, 8-bit column delta, value word; column of the first slot), pad a group's last chunk
//   5. the schedules      pieces (consecutive chunks of one range, at most one workgroup slot's share), largest first; shared accumulators
//                         and tickets of the ranges cut into several pieces
#include "block_utils.hpp"
#include "common.hpp"
#include "mask_probe.hpp"
#include "mxv_cob.hpp"
#include "mxv_cob_dyn.hpp"
#include "ops.hpp"
#include "plan.hpp"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <queue>
#include <vector>

using namespace splahip;

namespace {

    constexpr uint32_t kKeyTile = 2048;// entries per workgroup of the key pass
    constexpr unsigned kTpb     = 1024;// threads of a tile's workgroup

    // first index in a[0..n) with a[index] > x
    __device__ __forceinline__ uint32_t upper_bound_u32(const uint32_t* __restrict__ a, uint32_t n, uint32_t x) {
        uint32_t lo = 0, hi = n;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a[mid] <= x) lo = mid + 1;
            else
                hi = mid;
        }
        return lo;
    }

    constexpr uint32_t kSlotBits = 16;// slot field of a sort key, of the row map and of the slot stream: slots (and the 64 padding values R .. R + 63) below 0xFFFE

    __global__ __launch_bounds__(256) void cob_deg_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, uint32_t* __restrict__ deg, uint32_t* __restrict__ row_id) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < n_rows) {
            deg[i]    = Ap[i + 1] - Ap[i];
            row_id[i] = i;
        }
    }
    // deg_s is sorted descending: out[0] = number of rows of degree >= thr (one thread)
    __global__ void cob_count_ge_kernel(uint32_t n_rows, const uint32_t* __restrict__ deg_s, uint32_t thr, uint32_t* __restrict__ out) {
        uint32_t lo = 0, hi = n_rows;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (deg_s[mid] >= thr) lo = mid + 1;
            else
                hi = mid;
        }
        out[0] = lo;
    }
    // within a hub range the slots go in ROW order (the range's composition is by degree, its slot order is free): the rows' mask words and results are then
    // visited in ascending address order by the pass that finishes them
    __global__ __launch_bounds__(256) void cob_hub_key_kernel(uint32_t H, uint32_t R, const uint32_t* __restrict__ order, unsigned long long* __restrict__ key) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < H) key[i] = ((unsigned long long) (i / R) << 32) | order[i];
    }
    __global__ __launch_bounds__(256) void cob_hub_unkey_kernel(uint32_t H, const unsigned long long* __restrict__ key, uint32_t* __restrict__ order) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < H) order[i] = (uint32_t) key[i];
    }
    // the H rows of highest degree take hub slots, range by range in degree order: rowmap = range << 15 | slot, slot16 = hub; hub_row lists them (padded to whole ranges)
    __global__ __launch_bounds__(256) void cob_hub_kernel(uint32_t H, uint32_t H_padded, uint32_t R, const uint32_t* __restrict__ order, uint32_t* __restrict__ rowmap,
                                                          uint16_t* __restrict__ slot16, uint32_t* __restrict__ hub_row) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i >= H_padded) return;
        if (i < H) {
            const uint32_t row = order[i];
            rowmap[row]        = ((i / R) << kSlotBits) | (i % R);
            slot16[row]        = (uint16_t) kCobSlotHub;
            hub_row[i]         = row;
        } else
            hub_row[i] = 0xFFFFFFFFu;
    }
    // rows that take a natural slot (an entry, no hub) per group of 8 rows
    __global__ __launch_bounds__(256) void cob_group8_kernel(uint32_t n_rows, const uint32_t* __restrict__ Ap, const uint16_t* __restrict__ slot16, uint8_t* __restrict__ cnt8) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g * 8 >= n_rows) return;
        uint32_t c = 0;
        for (uint32_t row = g * 8; row < min(n_rows, g * 8 + 8); ++row) c += (Ap[row + 1] != Ap[row] && slot16[row] != kCobSlotHub) ? 1u : 0u;
        cnt8[g] = (uint8_t) c;
    }
    // natural slots: gmap[g] = range << 15 | slot of the group's first slot-taking row (from the host's greedy pass)
    __global__ __launch_bounds__(256) void cob_rowmap_kernel(uint32_t n_rows, uint32_t n_rows8, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ gmap,
                                                             uint32_t* __restrict__ rowmap, uint16_t* __restrict__ slot16) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g * 8 >= n_rows8) return;
        uint32_t m = gmap[g];
        for (uint32_t row = g * 8; row < g * 8 + 8; ++row) {
            if (row >= n_rows) {
                slot16[row] = (uint16_t) kCobSlotEmpty;// the padding behind the last row
                continue;
            }
            if (slot16[row] == kCobSlotHub) continue;
            if (Ap[row + 1] == Ap[row]) {
                slot16[row] = (uint16_t) kCobSlotEmpty;
                continue;
            }
            rowmap[row] = m;
            slot16[row] = (uint16_t) (m & ((1u << kSlotBits) - 1u));
            ++m;
        }
    }
    // fill (accumulators to the identity of the running op_add; tickets to 0)
    __global__ __launch_bounds__(256) void cob_fill32_kernel(size_t n, uint32_t* __restrict__ p, uint32_t w) {
        for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256) p[i] = w;
    }

    // ---- hot-column relabelling ----
    // references per column, of every `stride`-th entry (the atomics of the hub columns serialise: all 131 M entries of R-MAT 23 cost 10.8 ms, an eighth 1.4)
    __global__ __launch_bounds__(256) void cob_colcount_kernel(uint32_t nnz, const uint32_t* __restrict__ Aj, uint32_t* __restrict__ cnt, uint32_t stride) {
        for (uint64_t k = (uint64_t) (blockIdx.x * 256 + threadIdx.x) * stride; k < nnz; k += (uint64_t) gridDim.x * 256 * stride) atomicAdd(&cnt[Aj[k]], 1u);
    }
    // out[0] = columns referenced >= thr times, out[1] = those of them whose neighbour column (c ^ 1) is referenced >= thr times too, out[2..3] = the entries they hold (64 bit)
    __global__ __launch_bounds__(256) void cob_colstat_kernel(uint32_t n_cols, const uint32_t* __restrict__ cnt, uint32_t thr, uint32_t* __restrict__ out) {
        uint32_t           hot = 0, both = 0;
        unsigned long long held = 0;
        for (uint32_t c = blockIdx.x * 256 + threadIdx.x; c < n_cols; c += gridDim.x * 256)
            if (cnt[c] >= thr) {
                ++hot;
                held += cnt[c];
                if ((c ^ 1u) < n_cols && cnt[c ^ 1u] >= thr) ++both;
            }
        hot  = wave_sum(hot);
        both = wave_sum(both);
        for (unsigned off = kWave / 2; off > 0; off >>= 1) held += __shfl_xor(held, off, kWave);
        if (lane_id() == 0) {
            if (hot) atomicAdd(&out[0], hot);
            if (both) atomicAdd(&out[1], both);
            if (held) atomicAdd(reinterpret_cast<unsigned long long*>(out + 2), held);
        }
    }
    // the hot columns as (count, column) pairs for the sort; the others marked cold
    __global__ __launch_bounds__(256) void cob_hotlist_kernel(uint32_t n_cols, const uint32_t* __restrict__ cnt, uint32_t thr, uint32_t* __restrict__ n_out, uint32_t* __restrict__ key,
                                                              uint32_t* __restrict__ col, uint32_t* __restrict__ col_map) {
        const uint32_t c = blockIdx.x * 256 + threadIdx.x;
        const bool     hot = c < n_cols && cnt[c] >= thr;
        const unsigned long long m = __ballot(hot);
        uint32_t                 base = 0;
        if (lane_id() == 0 && m) base = atomicAdd(n_out, (uint32_t) __popcll(m));
        base = __shfl(base, 0, kWave);
        if (hot) {
            const uint32_t at = base + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
            key[at] = cnt[c], col[at] = c;
        } else if (c < n_cols)
            col_map[c] = 0xFFFFFFFFu;
    }
    __global__ __launch_bounds__(256) void cob_hotrank_kernel(uint32_t n_hot, const uint32_t* __restrict__ hot_cols, uint32_t* __restrict__ col_map) {
        const uint32_t i = blockIdx.x * 256 + threadIdx.x;
        if (i < n_hot) col_map[hot_cols[i]] = i;
    }
    // per call: the vector the kernels gather from = [v[hot_cols[0 .. K)] | zero padding to whole 64-word blocks | v]
    template<typename T>
    __global__ __launch_bounds__(256) void cob_vperm_kernel(uint32_t n_hot, uint32_t n_hot_pad, uint32_t n_cols, const uint32_t* __restrict__ hot_cols, const T* __restrict__ v,
                                                            T* __restrict__ out) {
        const uint32_t total = n_hot_pad + n_cols;
        for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) out[i] = i < n_hot ? v[hot_cols[i]] : (i < n_hot_pad ? T(0) : v[i - n_hot_pad]);
    }

    __global__ __launch_bounds__(256) void cob_key_kernel(uint32_t n_rows, uint32_t nnz, const uint32_t* __restrict__ Ap, const uint32_t* __restrict__ Aj,
                                                          const uint32_t* __restrict__ rowmap, uint32_t cbits, unsigned long long* __restrict__ key,
                                                          const uint32_t* __restrict__ col_map, uint32_t n_hot_pad) {
        const uint32_t k0 = blockIdx.x * kKeyTile;
        const uint32_t k1 = min(nnz, k0 + kKeyTile);
        // rows spanned by this tile of the entry stream (uniform searches: every lane reads the same words)
        const uint32_t r_lo = upper_bound_u32(Ap, n_rows + 1, k0) - 1;
        const uint32_t r_hi = upper_bound_u32(Ap, n_rows + 1, k1 - 1) - 1;
        for (uint32_t k = k0 + threadIdx.x; k < k1; k += 256) {
            const uint32_t row = r_lo + upper_bound_u32(Ap + r_lo, r_hi - r_lo + 1, k) - 1;
            const uint32_t m   = rowmap[row];
            const unsigned long long range = m >> kSlotBits;
            uint32_t c = Aj[k];
            if (col_map != nullptr) {// hot-column relabelling: a hot column takes its rank, every other one keeps its place behind the hot block
                const uint32_t h = col_map[c];
                c                = h != 0xFFFFFFFFu ? h : n_hot_pad + c;
            }
            key[k] = (((range << cbits) | c) << kSlotBits) | (m & ((1u << kSlotBits) - 1u));
        }
    }

    // start[g] = first sorted entry whose (range, window) group is >= g, g = 0 .. n_groups; chunks[g] = chunks of group g
    __global__ __launch_bounds__(256) void cob_group_kernel(uint32_t nnz, const unsigned long long* __restrict__ key, uint32_t shift, uint32_t n_groups,
                                                            uint32_t* __restrict__ start) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g > n_groups) return;
        uint32_t lo = 0, hi = nnz;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((key[mid] >> shift) < g) lo = mid + 1;
            else
                hi = mid;
        }
        start[g] = lo;
    }
    // Slots an entry takes in its group's chunks: itself, and — the column stream holds 8-bit DELTAS to the slot before — one bridging slot per 255 columns
    // of a gap of more than 255 to the entry before it (same group).  A bridging slot is a padding slot (a spare LDS word, value 0) at a column of its own.
    __global__ __launch_bounds__(256) void cob_weight_kernel(uint32_t nnz, const unsigned long long* __restrict__ key, uint32_t col_bits, uint32_t cbits, uint32_t* __restrict__ w) {
        const uint32_t k = blockIdx.x * 256 + threadIdx.x;
        if (k > nnz) return;
        if (k == nnz) {
            w[k] = 0u;
            return;
        }
        uint32_t x = 1u;
        if (k > 0 && (key[k] >> (kSlotBits + col_bits)) == (key[k - 1] >> (kSlotBits + col_bits))) {
            const uint32_t c = (uint32_t) (key[k] >> kSlotBits) & ((1u << cbits) - 1u), cp = (uint32_t) (key[k - 1] >> kSlotBits) & ((1u << cbits) - 1u);
            const uint32_t gap = c - cp;
            if (gap > 255u) x += (gap - 1u) / 255u;
        }
        w[k] = x;
    }
    __global__ __launch_bounds__(256) void cob_chunk_count_kernel(uint32_t n_groups, const uint32_t* __restrict__ start, const uint32_t* __restrict__ pos, uint32_t* __restrict__ chunks) {
        const uint32_t g = blockIdx.x * 256 + threadIdx.x;
        if (g < n_groups) chunks[g] = (pos[start[g + 1]] - pos[start[g]] + kCobChunk - 1) / kCobChunk;
    }

    // one workgroup (256 threads = 256 slots) per chunk.  The chunk's three streams: sl (16-bit slot in range; a padding / bridging slot carries R + lane),
    // dl (8-bit column delta to the slot before it in SORTED order; 0 for the chunk's first slot), ax (value word); hdr[chunk] = column of its first slot.
    __global__ __launch_bounds__(256) void cob_scatter_kernel(const unsigned long long* __restrict__ key, const uint32_t* __restrict__ val, const uint32_t* __restrict__ start,
                                                              const uint32_t* __restrict__ pos, const uint32_t* __restrict__ chunk_first, uint32_t n_groups, uint32_t R_hub, uint32_t R_nat,
                                                              uint32_t n_hub_ranges, uint32_t cbits, uint32_t win_bits, uint16_t* __restrict__ sl, uint8_t* __restrict__ dl,
                                                              uint32_t* __restrict__ ax, uint32_t* __restrict__ hdr, uint32_t* __restrict__ chunk_lines) {
        __shared__ uint32_t s_col[kCobChunk];
        const uint32_t c = blockIdx.x;
        const uint32_t g = upper_bound_u32(chunk_first, n_groups + 1, c) - 1;// largest g with chunk_first[g] <= c (uniform)
        const uint32_t i = c - chunk_first[g];
        const uint32_t s0 = start[g], s1 = start[g + 1];
        const uint32_t R  = (g >> win_bits) < n_hub_ranges ? R_hub : R_nat;// the group's range decides which slot value marks padding
        // memory position p = 4 * lane + j holds the chunk's sorted slot 64 * j + lane: a lane's loads bring its four slots, and the wave's j-th gather
        // instruction covers 64 consecutive sorted slots (ascending, neighbouring columns)
        const uint32_t si    = 64 * (threadIdx.x & 3) + (threadIdx.x >> 2);
        const uint32_t total = pos[s1] - pos[s0], S = i * kCobChunk + si, n_valid = min(kCobChunk, total - i * kCobChunk);
        uint32_t       slot = R + (threadIdx.x >> 2), x = 0, col = 0;// padding: the lane's spare LDS word (position 4 l + j belongs to lane l)
        if (S < total) {
            // the entry k whose slots cover S: pos[k] - pos[s0] <= S < pos[k + 1] - pos[s0]
            uint32_t lo = s0, hi = s1;
            while (hi - lo > 1) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (pos[mid] - pos[s0] <= S) lo = mid;
                else
                    hi = mid;
            }
            const unsigned long long kk = key[lo];
            const uint32_t           ck = (uint32_t) (kk >> kSlotBits) & ((1u << cbits) - 1u);
            const uint32_t           t  = S - (pos[lo] - pos[s0]), wk = pos[lo + 1] - pos[lo];
            if (t + 1 == wk) {// the entry itself (its bridging slots come first)
                slot = (uint32_t) kk & ((1u << kSlotBits) - 1u);
                x    = val[lo];
                col  = ck;
            } else
                col = ((uint32_t) (key[lo - 1] >> kSlotBits) & ((1u << cbits) - 1u)) + 255u * (t + 1u);// bridging slot t of the gap before the entry
        }
        s_col[si] = col;
        __syncthreads();
        if (S >= total) col = s_col[n_valid - 1];// padding behind the group's last slot: no further column
        __syncthreads();
        s_col[si] = col;
        __syncthreads();
        const uint32_t prev = si ? s_col[si - 1] : col;
        sl[(size_t) c * kCobChunk + threadIdx.x] = (uint16_t) slot;
        dl[(size_t) c * kCobChunk + threadIdx.x] = (uint8_t) (col - prev);
        ax[(size_t) c * kCobChunk + threadIdx.x] = x;
        // distinct 128-byte lines of v among the chunk's slots (they are sorted by column: count the changes): what the gathers of this chunk cost
        const bool fresh = si < n_valid && slot < R && (si == 0 || (col >> 5) != (prev >> 5));// (a padding / bridging slot gathers the chunk's base column: no line of its own)
        const int  lines = __syncthreads_count(fresh);
        if (threadIdx.x == 0) {
            hdr[c]         = s_col[0];
            chunk_lines[c] = (uint32_t) lines;
        }
    }

    __global__ __launch_bounds__(256) void cob_padchunk_kernel(uint32_t c, uint32_t R, uint16_t* __restrict__ sl, uint8_t* __restrict__ dl, uint32_t* __restrict__ ax,
                                                               uint32_t* __restrict__ hdr) {
        sl[(size_t) c * kCobChunk + threadIdx.x] = (uint16_t) (R + (threadIdx.x >> 2));
        dl[(size_t) c * kCobChunk + threadIdx.x] = 0;
        ax[(size_t) c * kCobChunk + threadIdx.x] = 0;
        if (threadIdx.x == 0) hdr[c] = 0;
    }

    // smallest and largest value word of the matrix (equal <=> one value, as in an adjacency matrix of ones)
    __global__ __launch_bounds__(256) void cob_ax_range_kernel(uint32_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        for (uint32_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
            const uint32_t w = x[k];
            lo = min(lo, w), hi = max(hi, w);
        }
        lo = ~wave_max(~lo);
        hi = wave_max(hi);
        // (one pair of atomics per WORKGROUP: same-address atomics serialise at ~35 ns each — a pair per wave of a 1024-workgroup grid cost this kernel 300 us)
        __shared__ uint32_t s_lo[256 / kWave], s_hi[256 / kWave];
        if (lane_id() == 0) s_lo[threadIdx.x / kWave] = lo, s_hi[threadIdx.x / kWave] = hi;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t k = 1; k < 256 / kWave; ++k) lo = min(lo, s_lo[k]), hi = max(hi, s_hi[k]);
            atomicMin(&out[0], lo);
            atomicMax(&out[1], hi);
        }
    }
    // per call, sparse-input-vector form: bit c of the bitmap = v[c] is not the value that cannot change a fold (0 for the zero family,
    // init — the step's "infinity" — for min-plus); a wave's ballot is two bitmap words
    template<typename T>
    __global__ __launch_bounds__(256) void cob_vbits_kernel(uint32_t n, const T* __restrict__ v, uint32_t* __restrict__ bits, uint32_t skip) {
        const uint32_t           c  = blockIdx.x * 256 + threadIdx.x;
        const unsigned long long b  = __ballot(c < n && to_bits(v[c]) != skip);
        if (lane_id() == 0) {
            bits[c >> 5]       = (uint32_t) b;
            bits[(c >> 5) + 1] = (uint32_t) (b >> 32);
        }
    }

    struct Scratch {
        std::vector<void*> ptrs;
        ~Scratch() {
            for (void* p : ptrs)
                if (p) (void) hipFree(p);
        }
        template<typename U>
        hipError_t alloc(U** out, size_t count) {
            void*      p = nullptr;
            hipError_t e = device_malloc(&p, sizeof(U) * (count ? count : 1));
            if (e == hipSuccess) ptrs.push_back(p);
            *out = static_cast<U*>(p);
            return e;
        }
    };

    uint32_t bits_for(uint64_t count) {// bits needed to tell `count` values apart (0 for count <= 1)
        uint32_t b = 0;
        while ((uint64_t(1) << b) < count) ++b;
        return b;
    }

    bool cob_disabled() {
        const char* e = knob_env("SPLAHIP_COB");
        return e && e[0] == '0';
    }
    uint32_t compute_units() {
        int             dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) return (uint32_t) prop.multiProcessorCount;
        return 256;
    }
    // LDS a workgroup may ask for on this device (the schedules are laid out for gfx950's 160 KB; a device with less is not served: CSR kernels)
    size_t lds_optin_bytes() {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, dev) == hipSuccess && v > 0) return (size_t) v;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) return (size_t) v;
        return 64 * 1024;
    }
    uint32_t env_u32(const char* name, uint32_t dflt) {
        const char* e = knob_env(name);
        return e && *e ? (uint32_t) std::strtoul(e, nullptr, 10) : dflt;
    }

    // set by cob_mxv_accumulate around its call (per host thread): the launch below then folds onto the values r already holds
    thread_local uint32_t tl_accumulate = 0;
    // the mask probe of the running call, waiting for a launch to ride on (cob_mxv_probed): the hub ranges' finishing kernel takes it as its first workgroups
    thread_local const MaskProbeArgs* tl_pending_probe = nullptr;

    // the mask probe as a launch of its own (mask_probe.hpp; a call served by the blocks)
    template<typename T>
    __global__ __launch_bounds__(256) void cob_mask_probe_kernel(MaskProbeArgs a, const uint32_t* __restrict__ Ap, const T* __restrict__ mask, int op_select) {
        mask_probe_body<T>(blockIdx.x, gridDim.x, a, Ap, mask, op_select);
    }

    // cob_hub_finish_kernel (mxv_cob.hpp) with the call's mask probe as its first workgroups: the probe is a hint for the NEXT masked call, and as a launch of its own
    // in front of the blocks it was 12 us of a ~290 us call
    template<typename T, int ACC, bool VXM>
    __global__ __launch_bounds__(256) void cob_hub_finish_probe_kernel(MaskProbeArgs probe, const uint32_t* __restrict__ Ap, uint32_t n_slots, uint32_t R, uint32_t acc_stride,
                                                                       uint32_t* __restrict__ acc, const uint32_t* __restrict__ hub_row, const T* __restrict__ mask, T* __restrict__ r,
                                                                       uint32_t* __restrict__ touched, uint32_t accumulate, DynBin add, int op_select, T init) {
        if (blockIdx.x < probe.blocks) {
            mask_probe_body<T>(blockIdx.x, probe.blocks, probe, Ap, mask, op_select);
            return;
        }
        const uint32_t i = (blockIdx.x - probe.blocks) * 256 + threadIdx.x;
        if (i >= n_slots) return;
        const size_t   ia  = (size_t) (i / R) * acc_stride + i % R;// accumulator q belongs to hub range q
        const uint32_t w   = acc[ia];
        const uint32_t row = hub_row[i];
        acc[ia]            = cob_identity<T, ACC>();
        if (row == 0xFFFFFFFFu) return;
        cob_hub_row<T, ACC, VXM>(w, row, mask, r, touched, accumulate, add, op_select, init);
    }

    static thread_local uint32_t tl_skip_bits = 0;// the v value whose entries the sparse-vector form skips in the running call (0; init for min-plus)

    // any negative / NaN value, and the largest value, of the copy's value words (padding slots hold 0)
    template<typename T>
    __global__ __launch_bounds__(256) void cob_ax_stats_kernel(size_t n, const T* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t bad = 0, mx = 0, nz = 0;
        for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t) gridDim.x * 256) {
            const T a = x[k];
            if constexpr (std::is_same<T, uint32_t>::value) mx = max(mx, a);
            else {
                if (a < T(0) || a != a) bad = 1;
                else if constexpr (std::is_same<T, int32_t>::value)
                    mx = max(mx, (uint32_t) a);
                if constexpr (std::is_same<T, float>::value) nz |= to_bits(a) == 0x80000000u ? 1u : 0u;
            }
        }
        bad = wave_max(bad);
        mx  = wave_max(mx);
        nz  = wave_max(nz);
        if (lane_id() == 0) {
            if (bad) atomicOr(&out[0], 1u);
            atomicMax(&out[1], mx);
            if (nz) atomicOr(&out[2], 1u);
        }
    }
    // does any value word of the copy read 0x80000000 (-0.0f as fp32)?  Asked once, at attach: the keyed fp32 MIN / MAX form needs to know (cob_float_minmax_ok)
    __global__ __launch_bounds__(256) void cob_negzero_kernel(size_t n, const uint32_t* __restrict__ x, uint32_t* __restrict__ out) {
        uint32_t nz = 0;
        for (size_t k = (size_t) blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t) gridDim.x * 256) nz |= x[k] == 0x80000000u ? 1u : 0u;
        nz = wave_max(nz);
        if (lane_id() == 0 && nz) atomicOr(out, 1u);// (only waves that saw one: rare)
    }
    template<typename T>
    int cob_ax_stats(const splahip_cob_s* g, hipStream_t st) {
        if (g->ax_type == TypeId<T>::value) return SPLAHIP_OK;
        DevMem out;
        SPLAHIP_TRY(out.alloc(st, 16));
        SPLAHIP_TRY(hipMemsetAsync(out.p, 0, 16, st));
        hipLaunchKernelGGL((cob_ax_stats_kernel<T>), dim3(1024), dim3(256), 0, st, (size_t) g->n_chunks * kCobChunk, reinterpret_cast<const T*>(g->d_ax), out.as<uint32_t>());
        uint32_t h[4] = {1, 0, 1, 0};
        SPLAHIP_TRY(hipMemcpyAsync(h, out.p, 16, hipMemcpyDeviceToHost, st));
        SPLAHIP_TRY(hipStreamSynchronize(st));
        g->ax_bad = h[0], g->ax_max = h[1], g->ax_type = TypeId<T>::value;
        return SPLAHIP_OK;
    }

    template<typename T, class Mult, int ACC, bool VXM, int MODE = 0>
    int launch_cob(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                   uint32_t init_bits, hipStream_t st) {
        constexpr int               W  = sizeof(typename CobSlice<T, ACC>::word) == 8 ? 1 : 0;
        const splahip_cob_sched_s&  sc = g->sched[W];
        (void) Ap;// which rows hold an entry is part of the copy (slot16)
        CobArgs                     a;
        a.tiles      = reinterpret_cast<const CobTile*>(sc.d_tiles);
        a.ranges     = reinterpret_cast<const CobRange*>(sc.d_ranges);
        a.hdr        = g->d_hdr;
        a.sl         = g->d_sl;
        a.dl         = g->d_dl;
        a.ax         = g->d_ax;
        a.slot16     = g->d_slot16;
        a.hub_row    = g->d_hub_row;
        a.acc        = g->d_acc;
        a.ticket     = g->d_ticket;
        a.fin        = g->d_ticket + (g->ticket_words - 32);
        a.touched    = touched;
        a.n_rows     = g->n_rows;
        a.R          = g->R;
        a.Rn         = g->Rn;
        a.n_hub_ranges = g->n_hub_ranges;
        a.acc_stride = g->acc_stride;
        a.col_bits   = g->col_bits;
        a.pad_chunk  = g->n_chunks;
        a.accumulate = VXM ? 0u : tl_accumulate;
        a.vbits      = g->d_vbits;
        a.x_uniform  = g->x_bits;
        if (g->n_hot) {// hot-column relabelling: the kernels gather from [v[hot columns] | v]
            hipLaunchKernelGGL((cob_vperm_kernel<T>), dim3(2048), dim3(256), 0, st, g->n_hot, g->n_hot_pad, g->n_cols, g->d_hot_cols, static_cast<const T*>(v),
                               reinterpret_cast<T*>(g->d_vperm));
            SPLAHIP_TRY(hipGetLastError());
            v = g->d_vperm;
        }
        if constexpr (MODE != 0) {// the bitmap of v for this call
            hipLaunchKernelGGL((cob_vbits_kernel<T>), dim3((g->n_vcols + 255) / 256), dim3(256), 0, st, g->n_vcols, static_cast<const T*>(v), g->d_vbits, tl_skip_bits);
            SPLAHIP_TRY(hipGetLastError());
        }
        const T init = from_bits<T>(init_bits);
        if (!sc.n_tiles) return fail(SPLAHIP_NOT_IMPLEMENTED, "the column-ordered blocks hold no schedule for this slice width");// (cob_serves says so beforehand)
        // the shared accumulators rest at the identity of the op_add that used them last: re-arm them when this call's differs
        const uint32_t n_multi_max = std::max(std::max(g->sched[0].n_multi, g->sched[1].n_multi), g->dyn ? g->n_ranges : 0u);
        if (n_multi_max && (!g->acc_armed || g->acc_identity != cob_identity<T, ACC>())) {
            hipLaunchKernelGGL(cob_fill32_kernel, dim3(256), dim3(256), 0, st, (size_t) n_multi_max * g->acc_stride, g->d_acc, cob_identity<T, ACC>());
            hipLaunchKernelGGL(cob_fill32_kernel, dim3(1), dim3(256), 0, st, (size_t) g->ticket_words, g->d_ticket, 0u);
            // the run-time schedule's words: cursors and states 0, next listed job = the grid (a workgroup's first job is its own index), nobody has left
            hipLaunchKernelGGL(cob_fill32_kernel, dim3(4), dim3(256), 0, st, (size_t) g->n_ranges * 3, g->d_dyn_words, 0u);
            for (int w2 = 0; w2 < 2; ++w2)
                if (g->sched[w2].d_dyn_ctl) {
                    hipLaunchKernelGGL(cob_fill32_kernel, dim3(1), dim3(64), 0, st, (size_t) 1, g->sched[w2].d_dyn_ctl, g->sched[w2].slots);
                    hipLaunchKernelGGL(cob_fill32_kernel, dim3(1), dim3(64), 0, st, (size_t) 1, g->sched[w2].d_dyn_ctl + 1, 0u);
                }
            SPLAHIP_TRY(hipGetLastError());
            g->acc_identity = cob_identity<T, ACC>();
            g->acc_armed    = true;
        }
        // a range of 16 Ki+ slots plus the spare words is over the 64 KB a kernel gets without asking (cheap; per device, so not cached in a static)
        // a double slice of at most half a CU's LDS: two 8-wave workgroups per CU instead of one 16-wave workgroup (their zeroing, pipeline fill and row
        // passes then run under each other's streams) — the fp32 PLUS forms only
        // (a launch that fails past this point may leave accumulators or tickets half used: the next call fills them again)
#define COB_TRY(expr)                                                       \
    do {                                                                    \
        const hipError_t cob_e_ = (expr);                                   \
        if (cob_e_ != hipSuccess) {                                         \
            g->acc_armed = false;                                           \
            return fail(SPLAHIP_ERROR, "%s failed: %s", #expr, hipGetErrorString(cob_e_)); \
        }                                                                   \
    } while (0)
        bool launched = false;
        if constexpr (W == 1 && MODE == 0) {
            if (g->half_groups) {
                constexpr unsigned kHalf = 512;
                COB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kHalf, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL((cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kHalf, MODE>), dim3(sc.n_tiles), dim3(kHalf), sc.lds_bytes, st, a,
                                   static_cast<const T*>(v), static_cast<const T*>(mask), static_cast<T*>(r), Mult(op_mult), DynBin(op_add), op_select, init);
                COB_TRY(hipGetLastError());
                launched = true;
            }
        }
        if (!launched && g->dyn && sc.n_jobs) {
            CobDynArgs d;
            d.jobs          = sc.d_jobs;
            d.spans         = reinterpret_cast<const CobDynSpan*>(g->d_spans);
            d.cursor        = g->d_dyn_words;
            d.state         = g->d_dyn_words + g->n_ranges;
            d.ctl           = sc.d_dyn_ctl;
            d.n_jobs        = sc.n_jobs;
            d.n_ranges      = g->n_ranges;
            d.weight        = g->d_dyn_weight;
            d.fin_flag      = g->d_dyn_words + 2 * (size_t) g->n_ranges;
            g->dyn_seq      = g->dyn_seq + 1 ? g->dyn_seq + 1 : 1;
            d.seq           = g->dyn_seq;
            a.ranges        = reinterpret_cast<const CobRange*>(g->d_dyn_ranges);
            d.min_cost_hub   = g->dyn_min_hub;
            d.min_cost_nat   = g->dyn_min_nat;
            COB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cob_dyn_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 128));
            hipLaunchKernelGGL((cob_dyn_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), dim3(sc.slots), dim3(kTpb), sc.lds_bytes, st, a, d, static_cast<const T*>(v),
                               static_cast<const T*>(mask), static_cast<T*>(r), Mult(op_mult), DynBin(op_add), op_select, init);
            COB_TRY(hipGetLastError());
            // the rows of every range: hub slots and natural row tasks in one launch
            const uint32_t n_slots = g->n_hub_ranges * g->R, hub_blocks = (n_slots + 255) / 256;
            if (hub_blocks + g->n_fin_tasks) {
                hipLaunchKernelGGL((cob_finish_kernel<T, ACC, VXM>), dim3(hub_blocks + g->n_fin_tasks), dim3(256), 0, st, hub_blocks, n_slots, g->R, g->acc_stride, g->d_acc, g->d_hub_row,
                                   reinterpret_cast<const CobRange*>(g->d_fin_tasks), d.fin_flag, d.seq, g->d_slot16, g->n_rows, static_cast<const T*>(mask), static_cast<T*>(r), touched, a.accumulate,
                                   DynBin(op_add), op_select, init);
                COB_TRY(hipGetLastError());
            }
            return SPLAHIP_OK;
        }
        if (!launched) {
            COB_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL((cob_mxv_kernel<T, Mult, DynBin, ACC, VXM, kTpb, MODE>), dim3(sc.n_tiles), dim3(kTpb), sc.lds_bytes, st, a,
                               static_cast<const T*>(v), static_cast<const T*>(mask), static_cast<T*>(r), Mult(op_mult), DynBin(op_add), op_select, init);
            COB_TRY(hipGetLastError());
        }
        if (g->hub_finish == 1 && g->n_hub_ranges) {
            const uint32_t n_slots = g->n_hub_ranges * g->R;
            const MaskProbeArgs* probe = tl_pending_probe;
            if (probe && probe->blocks && mask != nullptr) {
                tl_pending_probe = nullptr;// (taken)
                hipLaunchKernelGGL((cob_hub_finish_probe_kernel<T, ACC, VXM>), dim3(probe->blocks + (n_slots + 255) / 256), dim3(256), 0, st, *probe, Ap, n_slots, g->R, g->acc_stride,
                                   g->d_acc, g->d_hub_row, static_cast<const T*>(mask), static_cast<T*>(r), touched, a.accumulate, DynBin(op_add), op_select, init);
            } else
                hipLaunchKernelGGL((cob_hub_finish_kernel<T, ACC, VXM>), dim3((n_slots + 255) / 256), dim3(256), 0, st, n_slots, g->R, g->acc_stride, g->d_acc, g->d_hub_row,
                                   static_cast<const T*>(mask), static_cast<T*>(r), touched, a.accumulate, DynBin(op_add), op_select, init);
            COB_TRY(hipGetLastError());
        }
#undef COB_TRY
        return SPLAHIP_OK;
    }

    template<typename T, int ACC, bool VXM>
    int cob_by_mult(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                    uint32_t init_bits, hipStream_t st) {
        // the sparse-input-vector form (mxv_cob.hpp MODE 1 / 2) for the semirings in which a zero of v cannot change a row (sell_skip_family == 1): int / uint only
        if constexpr (!VXM && !std::is_same<T, float>::value && (ACC == COB_OR || ACC == COB_ADD || ACC == COB_XOR || ACC == COB_MAX)) {
            const char* be = knob_env("SPLAHIP_COB_BITS");
            if (g->d_vbits && !(be && be[0] == '0') && !tl_accumulate && sell_skip_family(TypeId<T>::value, op_mult, op_add) == 1) {
                const char* ue = knob_env("SPLAHIP_COB_UNIFORM");
                if (g->x_uniform && !(ue && ue[0] == '0')) {
                    if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM, 2>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                    return launch_cob<T, DynBin, ACC, VXM, 2>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                }
                if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                return launch_cob<T, DynBin, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            }
        }
        // the same form for min-plus (PLUS / MIN on int / uint, sell_skip_family == 2: the shortest-path step): an entry whose v value is `init` cannot
        // lower a minimum that starts at init, provided no matrix value is negative and init + max(A) does not wrap (checked once per type / per call)
        if constexpr (!VXM && ACC == COB_MIN) {
            const char* be = knob_env("SPLAHIP_COB_BITS");
            // (worth it once v no longer fits the L2s: R-MAT 24, 67 MB of v: 0.83 -> 0.70 ms; R-MAT 22, 16.8 MB: 178 -> 204 us.  SPLAHIP_COB_BITS=2 forces it)
            const bool big = g->n_vcols >= (1u << 23) || (be && be[0] == '2');
            if (g->d_vbits && big && !(be && be[0] == '0') && !tl_accumulate && sell_skip_family(TypeId<T>::value, op_mult, op_add) == 2) {
                if (int rc = cob_ax_stats<T>(g, st)) return rc;
                // (fp32: x >= 0 gives x + init >= init for every init that is no NaN — the reference's fp32 SSSP step, init = +inf)
                const bool fits = std::is_same<T, float>::value ? from_bits<float>(init_bits) == from_bits<float>(init_bits)
                                  : std::is_same<T, int32_t>::value ? (long long) (int32_t) init_bits + (long long) g->ax_max <= 0x7FFFFFFFll
                                                                    : (unsigned long long) init_bits + (unsigned long long) g->ax_max <= 0xFFFFFFFFull;
                if (!g->ax_bad && fits) {
                    tl_skip_bits = init_bits;
                    const int rc = launch_cob<T, StaBin<SPLAHIP_PLUS>, ACC, VXM, 1>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                    tl_skip_bits = 0;
                    return rc;
                }
            }
        }
        // the semirings of the configs get a compile-time op_mult; every other pair runs the runtime-switched functor
        if constexpr (ACC == COB_ADD) {
            if (op_mult == SPLAHIP_MULT) return launch_cob<T, StaBin<SPLAHIP_MULT>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_MIN || (ACC == COB_MAX && std::is_same<T, float>::value)) {
            if (op_mult == SPLAHIP_PLUS) return launch_cob<T, StaBin<SPLAHIP_PLUS>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_OR && !std::is_same<T, float>::value) {
            if (op_mult == SPLAHIP_BAND) return launch_cob<T, StaBin<SPLAHIP_BAND>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        if constexpr (ACC == COB_LOR) {
            if (op_mult == SPLAHIP_LAND) return launch_cob<T, StaBin<SPLAHIP_LAND>, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        return launch_cob<T, DynBin, ACC, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
    }

    template<typename T, bool VXM>
    int cob_typed(const splahip_cob_s* g, const uint32_t* Ap, const void* mask, const void* v, void* r, uint32_t* touched, int op_mult, int op_add, int op_select,
                  uint32_t init_bits, hipStream_t st) {
        constexpr int type = TypeId<T>::value;
        switch (cob_acc_kind(op_add, type)) {
            case COB_ADD: return cob_by_mult<T, COB_ADD, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            case COB_LOR: return cob_by_mult<T, COB_LOR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            case COB_LAND: return cob_by_mult<T, COB_LAND, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            default: break;
        }
        if constexpr (!std::is_same<T, float>::value) {
            switch (cob_acc_kind(op_add, type)) {
                case COB_MIN: return cob_by_mult<T, COB_MIN, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_MAX: return cob_by_mult<T, COB_MAX, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_OR: return cob_by_mult<T, COB_OR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_AND: return cob_by_mult<T, COB_AND, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                case COB_XOR: return cob_by_mult<T, COB_XOR, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
                default: break;
            }
        } else {// fp32 MIN / MAX over op_mult = PLUS: the keyed slice (mxv_cob.hpp: cob_fkey); cob_float_minmax_ok has been asked by the caller
            if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MIN) return cob_by_mult<T, COB_MIN, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
            if (op_mult == SPLAHIP_PLUS && op_add == SPLAHIP_MAX) return cob_by_mult<T, COB_MAX, VXM>(g, Ap, mask, v, r, touched, op_mult, op_add, op_select, init_bits, st);
        }
        return fail(SPLAHIP_NOT_IMPLEMENTED, "the column-ordered-blocks form has no LDS atomic for this op_add");
    }

    // fp32 MIN / MAX over PLUS: served when the copy's values hold no -0.0f (then no product is -0.0f; mxv_cob.hpp: cob_fkey) — known since the attach
    bool cob_float_minmax_ok(const splahip_cob_s* g) { return g->ax_negzero == 0; }

}// namespace

namespace splahip {

    void cob_release(splahip_cob_s* g) {
        if (!g) return;
        if (g->h_probe) (void) hipHostFree(const_cast<uint32_t*>(g->h_probe));
        for (void* p : {(void*) g->d_hdr, (void*) g->d_sl, (void*) g->d_dl, (void*) g->d_ax, (void*) g->d_slot16, (void*) g->d_hub_row, (void*) g->d_acc, (void*) g->d_ticket, (void*) g->d_vbits, (void*) g->d_probe, (void*) g->d_hot_cols, (void*) g->d_vperm,
                        (void*) g->sched[0].d_tiles, (void*) g->sched[0].d_ranges, (void*) g->sched[1].d_tiles, (void*) g->sched[1].d_ranges, (void*) g->d_spans, (void*) g->d_dyn_words,
                        (void*) g->sched[0].d_jobs, (void*) g->d_fin_tasks, (void*) g->d_dyn_weight, (void*) g->d_dyn_ranges, (void*) g->sched[0].d_dyn_ctl, (void*) g->sched[1].d_jobs,
                        (void*) g->sched[1].d_dyn_ctl})
            if (p) (void) hipFree(p);
        delete g;
    }

    bool cob_serves(const splahip_plan_s* plan, int type, int op_add, uint32_t n_cols) {
        if (!plan || !plan->cob || plan->cob->n_cols != n_cols || cob_disabled()) return false;
        const int acc = cob_acc_kind(op_add, type);
        if (acc == COB_NONE) return false;
        const bool wide = acc == COB_ADD && type == SPLAHIP_FLOAT;// fp32 sums keep a double slice
        if (wide && deterministic_calls()) return false;            // partial sums meet in arrival order: the fixed-order kernels serve a deterministic call
        return plan->cob->sched[wide ? 1 : 0].n_tiles != 0;
    }

    // the masked SpMV's own question: cob_serves, plus fp32 MIN / MAX over op_mult = PLUS (the reference's fp32 SSSP semiring) when the matrix
    // holds no -0.0f (the attach looked).  No device work: a query (splahip_mxv_variant_of) may ask too
    bool cob_serves_mxv(const splahip_plan_s* plan, int type, int op_mult, int op_add, uint32_t n_cols, void* stream) {
        if (cob_serves(plan, type, op_add, n_cols)) return true;
        if (!plan || !plan->cob || plan->cob->n_cols != n_cols || cob_disabled()) return false;
        if (type != SPLAHIP_FLOAT || (op_add != SPLAHIP_MIN && op_add != SPLAHIP_MAX) || cob_acc_kind_mxv(op_add, op_mult, type) == COB_NONE || plan->cob->sched[0].n_tiles == 0)
            return false;
        const char* e = knob_env("SPLAHIP_COB_FMINMAX");
        if (e && e[0] == '0') return false;
        (void) stream;
        return cob_float_minmax_ok(plan->cob);
    }

    int cob_mxv(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                uint32_t init_bits, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return cob_typed<int32_t, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_UINT: return cob_typed<uint32_t, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
            case SPLAHIP_FLOAT: return cob_typed<float, false>(plan->cob, Ap, mask, v, r, nullptr, op_mult, op_add, op_select, init_bits, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

    // A masked call over a plan that holds the blocks: which side serves it (1: the selected-rows kernels) — the verdict of the latest probe that has FINISHED, of an
    // earlier call: the masks of consecutive calls usually select alike — and the probe of THIS call's mask (out->blocks == 0: none).  The side that serves the call
    // launches it: cob_mask_probe_launch in front of the blocks, or the first workgroups of the row kernels' first launch (mxv.hip: launch_rows).
    int cob_mask_probe_prepare(const splahip_plan_s* plan, int op_select, MaskProbeArgs* out) {
        *out                   = MaskProbeArgs();
        const splahip_cob_s* g = plan ? plan->cob : nullptr;
        if (!g || !g->d_probe || !g->h_probe || op_select == SPLAHIP_ALWAYS || env_u32("SPLAHIP_MASK_PROBE", 1) == 0) return 0;
        const int      rows   = g->h_probe[3] != 0u && g->h_probe[2] != 0u ? 1 : 0;
        const uint32_t stride = std::max(1u, std::min(64u, g->n_rows >> 16));// a 1-in-64 sample once there are millions of rows; every row of a small matrix
        // a selected entry costs the row kernels ~4.2 x what an entry costs the blocks, whatever its row's length (round 6, R-MAT 23: 9.5 us per million selected
        // entries — the chip's ceiling of ~110 G four-byte gathers a second — on top of ~60 us of passes over mask / Ap / r, against 2.25 ns per stored entry on the
        // blocks: the row kernels win below ~18 % of the entries.  Until round 6 a medium row was one lane's and the heaviest rows went through the wave-tile kernel:
        // 9 x and 30 x, and a random 10 % of all rows stayed on the blocks at 1.09 x the all-rows call; now 0.6 x)
        const uint32_t groups = (g->n_rows + stride * 2048u - 1) / (stride * 2048u);
        g->probe_seq          = g->probe_seq + 1 ? g->probe_seq + 1 : 1;
        out->blocks           = std::min(groups, 1024u);
        out->n_rows = g->n_rows, out->stride = stride, out->rows_x10 = env_u32("SPLAHIP_MASK_ROWS_X10", 45), out->long_x10 = env_u32("SPLAHIP_MASK_LONG_X10", 45);
        out->nnz = plan->nnz, out->seq = g->probe_seq, out->probe = g->d_probe, out->host_words = g->d_h_probe, out->hub_row = g->d_hub_row;
        out->heavy_slots = stride > 1 ? g->heavy_slots : 0u, out->heavy_deg = stride > 1 ? g->heavy_deg : 0xFFFFFFFFu;
        g->last_rows = rows;
        return rows;
    }
    void cob_mask_probe_launch(const MaskProbeArgs& a, int type, const uint32_t* Ap, const void* mask, int op_select, void* stream) {
        if (a.blocks == 0) return;
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: hipLaunchKernelGGL((cob_mask_probe_kernel<int32_t>), dim3(a.blocks), dim3(256), 0, st, a, Ap, static_cast<const int32_t*>(mask), op_select); break;
            case SPLAHIP_UINT: hipLaunchKernelGGL((cob_mask_probe_kernel<uint32_t>), dim3(a.blocks), dim3(256), 0, st, a, Ap, static_cast<const uint32_t*>(mask), op_select); break;
            default: hipLaunchKernelGGL((cob_mask_probe_kernel<float>), dim3(a.blocks), dim3(256), 0, st, a, Ap, static_cast<const float*>(mask), op_select); break;
        }
        (void) hipGetLastError();// (a probe that did not launch leaves the hint as it was)
    }

    // a masked call served by the blocks, with its probe: the probe rides on the hub ranges' finishing launch when there is one (SPLAHIP_MASK_PROBE_FUSED=0: never),
    // else it is a launch of its own behind the product — either way its verdict is there long before the next call asks
    int cob_mxv_probed(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add, int op_select,
                       uint32_t init_bits, const MaskProbeArgs& probe, void* stream) {
        tl_pending_probe = probe.blocks && env_u32("SPLAHIP_MASK_PROBE_FUSED", 1) ? &probe : nullptr;
        const bool rode  = tl_pending_probe != nullptr;
        if (!rode) cob_mask_probe_launch(probe, type, Ap, mask, op_select, stream);
        const int st = cob_mxv(plan, type, Ap, mask, v, r, op_mult, op_add, op_select, init_bits, stream);
        if (tl_pending_probe) {// nobody took it
            tl_pending_probe = nullptr;
            cob_mask_probe_launch(probe, type, Ap, mask, op_select, stream);
        }
        return st;
    }

    // second column part of a row block (dist.hip): r[row] = add(r[row], fold of this part's products) for the selected rows that hold an entry here
    int cob_mxv_accumulate(const splahip_plan_s* plan, int type, const uint32_t* Ap, const void* mask, const void* v, void* r, int op_mult, int op_add,
                           int op_select, void* stream) {
        tl_accumulate = 1;
        const int st  = cob_mxv(plan, type, Ap, mask, v, r, op_mult, op_add, op_select, 0u, stream);
        tl_accumulate = 0;
        return st;
    }

    // the pull form of vxm with every index of v stored, over the copy attached to the TRANSPOSED matrix's plan: out values and touched flags per
    // output index.  Only an op_add with a true identity (not LOR / LAND: a lone product must come out as itself, not as its truth value)
    bool cob_vxm_serves(const splahip_plan_s* plan_T, int type, int op_add, uint32_t n_in) {
        return op_add != SPLAHIP_LOR && op_add != SPLAHIP_LAND && cob_serves(plan_T, type, op_add, n_in);
    }
    int cob_vxm(const splahip_plan_s* plan_T, int type, const uint32_t* Tp, const void* mask, const void* v, void* out, uint32_t* touched, int op_mult, int op_add,
                int op_select, void* stream) {
        hipStream_t st = as_stream(stream);
        switch (type) {
            case SPLAHIP_INT: return cob_typed<int32_t, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
            case SPLAHIP_UINT: return cob_typed<uint32_t, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
            case SPLAHIP_FLOAT: return cob_typed<float, true>(plan_T->cob, Tp, mask, v, out, touched, op_mult, op_add, op_select, 0u, st);
        }
        return fail(SPLAHIP_INVALID_ARGUMENT, "unknown element type");
    }

}// namespace splahip

namespace {
    // library default for the natural ranges' slot count (see splahip_csr_plan_attach_cob)
    uint32_t nat_slots_default(uint32_t R, bool explicit_bits) {
        if (explicit_bits || R != 20224u) return R;// a caller that sets the geometry gets one size
        return 32704u;// (the library's own geometry, where the attach finds that it pays — see there.  The slot field allows up to 40832 = 160 KB of float words
                      // (SPLAHIP_COB_NAT_SLOTS); measured no better: R-MAT 23 278 against 271 us, R-MAT 24 611 against 609)
    }
}// namespace

extern "C" int splahip_csr_plan_attach_cob(splahip_plan plan, uint32_t n_cols, const uint32_t* d_Ap, const uint32_t* d_Aj, const void* d_Ax, uint32_t row_bits,
                                           uint32_t tile_chunks, void* stream) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (plan->cob) {
        cob_release(plan->cob);
        plan->cob = nullptr;
    }
    if (plan->nnz == 0 || plan->n_rows == 0 || n_cols == 0) return SPLAHIP_OK;// nothing to lay out: mxv keeps the CSR kernels
    SPLAHIP_REQUIRE(d_Ap && d_Aj && d_Ax, "null device pointer");
    // slots per range: row_bits = 0 -> the library's choice, 20224 (a double slice of that many rows + the spare words fills the 160 KB of LDS a CU has,
    // and two 4-byte slices fit it together); row_bits in [8, 14] -> 2^row_bits (tests: many ranges and pieces on small inputs)
    uint32_t R = row_bits ? 0u : env_u32("SPLAHIP_COB_SLOTS", 0);
    if (R == 0) {
        if (row_bits == 0) row_bits = env_u32("SPLAHIP_COB_ROW_BITS", 0);
        SPLAHIP_REQUIRE(row_bits == 0 || (row_bits >= 8 && row_bits <= 14), "row_bits must be 0 or in [8, 14] (the range's result slice lives in LDS)");
        R = row_bits ? (1u << row_bits) : 20224u;
    }
    SPLAHIP_REQUIRE(R % 64 == 0 && R >= 256 && R <= 20224, "slots per range must be a multiple of 64 in [256, 20224]");
    // slots of a NATURAL range (the low-degree rows, in row order): at least R.  More rows per range = fewer ranges = fewer (range, line of v) pairs, and
    // the natural ranges are where a scale-free matrix's gather traffic is (profiles/r03_notes.md 6).  A natural range's fp32 sum then lives in FLOAT
    // words (compare-and-swap adds: its rows have few entries each, so neither the adds collide nor the sum needs the double), 32704 of them = 128 KB;
    // 4-byte slices of that size run one workgroup per CU.  SPLAHIP_COB_NAT_SLOTS overrides (tests: 2 x R on small inputs).
    const bool nat_auto = knob_env("SPLAHIP_COB_NAT_SLOTS") == nullptr;// the size is the library's proposal: the attach may take it back (below)
    uint32_t   Rn       = env_u32("SPLAHIP_COB_NAT_SLOTS", nat_slots_default(R, row_bits != 0));
    if (Rn < R) Rn = R;
    SPLAHIP_REQUIRE(Rn % 64 == 0 && Rn <= 40832, "natural slots per range must be a multiple of 64, at most 40832");
    if (tile_chunks == 0) tile_chunks = env_u32("SPLAHIP_COB_TILE_CHUNKS", 0);// 0: pieces are cut at range and share boundaries only
    int st = ensure_device();
    if (st != SPLAHIP_OK) return st;
    hipStream_t    s      = as_stream(stream);
    const uint32_t n_rows = plan->n_rows, nnz = plan->nnz;
    const uint32_t n_rows8 = (n_rows + 7) & ~7u;

    const uint32_t col_bits = 16;                                        // a chunk never straddles a window of 2^16 columns: its decoded words are slot << 16 | column offset

    // (SPLAHIP_COB_TRACE: wall time of the attach's phases on stderr; each stamp synchronises the stream)
    const bool trace_phases = knob_env("SPLAHIP_COB_TRACE") != nullptr;
    auto       t_last       = std::chrono::steady_clock::now();
    auto       stamp        = [&](const char* what) {
        if (!trace_phases) return;
        (void) hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cob] attach phase %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    Scratch tmp;
    auto*   g   = new splahip_cob_s();
    g->n_rows   = n_rows;
    g->n_cols   = n_cols;
    g->n_vcols  = n_cols;
    // Hot-column relabelling.  What a range's gathers cost is the number of distinct 128-byte lines of v it touches; in a matrix whose column labels carry no
    // locality (a real graph's; `--permute 1`) a hot column shares its line with 31 cold ones and every range fetches the line for it (profiles/r04_notes.md 1:
    // R-MAT 23, 14.8 M (range, line) pairs with random labels against 7.3 M with the generator's).  The columns referenced >= max(16, average) times take new ids 0 .. K-1 in
    // descending reference-count order — hot columns side by side, the hottest lines shared by every range — and a call gathers from [v[hot columns] | v] (one
    // pass per call, ~15 us at R-MAT 23).  Decided here, per matrix: only where the labels do NOT already cluster the hot columns (a hot column's neighbour c ^ 1
    // is mostly hot in a generator's labelling, rarely in a random one).  SPLAHIP_COB_RELABEL = 0 never, 2 always.
    uint32_t* col_map = nullptr;
    {
        const uint32_t mode = env_u32("SPLAHIP_COB_RELABEL", 1);
        if (mode != 0 && n_cols >= (1u << 16) && (uint64_t) n_cols + n_cols / 2 < (1ull << 31)) {
            uint32_t *cnt = nullptr, *st4 = nullptr;
            hipError_t e0 = tmp.alloc(&cnt, n_cols);
            if (e0 == hipSuccess) e0 = tmp.alloc(&st4, 4);
            if (e0 == hipSuccess) e0 = hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (size_t) n_cols, s);
            if (e0 == hipSuccess) e0 = hipMemsetAsync(st4, 0, 16, s);
            uint32_t       h4[4]   = {0, 0, 0, 0};
            const uint32_t hot_thr = env_u32("SPLAHIP_COB_RELABEL_MIN", std::max<uint32_t>(16, nnz / n_cols));// (hot: referenced at least 16 times and at least as often as the average column)
            // the decision needs statistics, not counts: first an eighth of the entries against an eighth of the threshold; the exact counts (they order the hot
            // columns) only for a matrix the sample says will be relabelled
            auto count_pass = [&](uint32_t stride, uint32_t thr) {
                hipLaunchKernelGGL(cob_colcount_kernel, dim3(4096), dim3(256), 0, s, nnz, d_Aj, cnt, stride);
                hipLaunchKernelGGL(cob_colstat_kernel, dim3(1024), dim3(256), 0, s, n_cols, cnt, thr, st4);
                hipError_t e1 = hipMemcpyAsync(h4, st4, 16, hipMemcpyDeviceToHost, s);
                if (e1 == hipSuccess) e1 = hipStreamSynchronize(s);
                return e1;
            };
            const uint32_t sample = mode == 2 || nnz < (1u << 24) ? 1u : 8u;
            if (e0 == hipSuccess) e0 = count_pass(sample, std::max(2u, hot_thr / sample));
            if (e0 == hipSuccess && sample != 1u) {
                const uint64_t held_s = (((uint64_t) h4[3] << 32) | h4[2]) * sample;
                const bool     likely = h4[0] != 0 && (double) h4[1] < 0.6 * h4[0] && (double) h4[0] <= 0.2 * n_cols && (double) held_s >= 0.5 * nnz;// (a little wider than the rule below)
                if (likely) {
                    e0 = hipMemsetAsync(cnt, 0, sizeof(uint32_t) * (size_t) n_cols, s);
                    if (e0 == hipSuccess) e0 = hipMemsetAsync(st4, 0, 16, s);
                    if (e0 == hipSuccess) e0 = count_pass(1u, hot_thr);
                } else
                    h4[0] = h4[1] = h4[2] = h4[3] = 0;// not a candidate: no hot list
            }
            const uint32_t K = h4[0];
            const bool     scattered = K != 0 && (double) h4[1] < 0.5 * K;// fewer than half of the hot columns sit beside another hot column
            // ... and the matrix must be SKEWED: few hot columns holding most entries (R-MAT 23: 10 % of the columns, 90 % of the entries).  On a flatter degree
            // sequence (the com-Orkut stand-in: a quarter of the columns above the average) every range touches nearly every line either way, and the pass over v
            // per call is all the relabelling adds: 632 us against 579 on the PageRank step
            const uint64_t held   = ((uint64_t) h4[3] << 32) | h4[2];
            const bool     skewed = (double) K <= 0.15 * n_cols && (double) held >= 0.6 * nnz;
            if (e0 == hipSuccess && K >= (mode == 2 ? 64u : 1024u) && K <= n_cols / 2 && (mode == 2 || (scattered && skewed))) {
                uint32_t *hk = nullptr, *hc = nullptr, *hk_s = nullptr;
                e0 = tmp.alloc(&hk, K);
                if (e0 == hipSuccess) e0 = tmp.alloc(&hc, K);
                if (e0 == hipSuccess) e0 = tmp.alloc(&hk_s, K);
                if (e0 == hipSuccess) e0 = tmp.alloc(&col_map, n_cols);
                if (e0 == hipSuccess) e0 = device_malloc(reinterpret_cast<void**>(&g->d_hot_cols), sizeof(uint32_t) * (size_t) K);
                if (e0 == hipSuccess) {
                    g->bytes += sizeof(uint32_t) * (size_t) K;
                    e0 = hipMemsetAsync(st4, 0, 16, s);
                }
                size_t bytes = 0;
                if (e0 == hipSuccess) {
                    hipLaunchKernelGGL(cob_hotlist_kernel, dim3((n_cols + 255) / 256), dim3(256), 0, s, n_cols, cnt, hot_thr, st4, hk, hc, col_map);
                    e0 = rocprim::radix_sort_pairs_desc(nullptr, bytes, hk, hk_s, hc, g->d_hot_cols, (size_t) K, 0, 32, s);
                }
                void* st_tmp = nullptr;
                if (e0 == hipSuccess) e0 = device_malloc(&st_tmp, bytes ? bytes : 4);
                if (e0 == hipSuccess) {
                    tmp.ptrs.push_back(st_tmp);
                    e0 = rocprim::radix_sort_pairs_desc(st_tmp, bytes, hk, hk_s, hc, g->d_hot_cols, (size_t) K, 0, 32, s);
                }
                if (e0 == hipSuccess) {
                    hipLaunchKernelGGL(cob_hotrank_kernel, dim3((K + 255) / 256), dim3(256), 0, s, K, g->d_hot_cols, col_map);
                    e0 = hipGetLastError();
                }
                if (e0 == hipSuccess) {
                    g->n_hot     = K;
                    g->n_hot_pad = (K + 63u) & ~63u;
                    g->n_vcols   = g->n_hot_pad + n_cols;
                    e0           = device_malloc(reinterpret_cast<void**>(&g->d_vperm), sizeof(uint32_t) * (size_t) g->n_vcols);
                    if (e0 == hipSuccess) g->bytes += sizeof(uint32_t) * (size_t) g->n_vcols;
                }
            }
            if (e0 != hipSuccess) {
                cob_release(g);
                return fail(SPLAHIP_ERROR, "column-ordered-blocks plan failed (column relabelling): %s", hipGetErrorString(e0));
            }
            if (!g->n_hot) col_map = nullptr;
            if (knob_env("SPLAHIP_COB_TRACE"))
                std::fprintf(stderr, "[cob] columns referenced >= %u times: %u, %u of them beside another one: relabelling %s\n", hot_thr, K, h4[1], g->n_hot ? "ON" : "off");
        }
    }
    const uint32_t cbits    = std::max(bits_for(g->n_vcols), col_bits);  // column field of the sort key: at least one whole window
    const uint32_t win_bits = cbits - col_bits;                          // column windows per range = 2^win_bits
    g->R        = R;
    g->col_bits = col_bits;
    // the hub ranges' row pass as a kernel of its own behind the main launch (mxv_cob.hpp: cob_hub_finish_kernel).  Measured with it / without: fp32 plus-times
    // R-MAT 22 154 - 156 / 162 - 166 us, R-MAT 23 322 - 330 / 321 - 331 us; int32 min-plus 189 / 195 and 302 / 319 us.  SPLAHIP_COB_HUB_FINISH=0: one launch
    g->hub_finish  = std::min(1u, env_u32("SPLAHIP_COB_HUB_FINISH", 1));
    hipError_t e    = hipSuccess;
    auto       bail = [&](const char* what) {
        cob_release(g);
        return fail(SPLAHIP_ERROR, "column-ordered-blocks plan failed (%s): %s", what, hipGetErrorString(e));
    };
    auto own = [&](auto** q, size_t count) {
        using U = std::remove_pointer_t<std::remove_pointer_t<decltype(q)>>;
        if (e == hipSuccess) e = device_malloc(reinterpret_cast<void**>(q), sizeof(U) * (count ? count : 1));
        if (e == hipSuccess) g->bytes += sizeof(U) * (count ? count : 1);
    };

    stamp("column statistics / relabel");
    // 0: rows -> (range, slot).  Hub rows first.
    uint32_t *deg = nullptr, *deg_s = nullptr, *row_id = nullptr, *order = nullptr, *rowmap = nullptr, *cnt = nullptr;
    e = tmp.alloc(&deg, n_rows);
    if (e == hipSuccess) e = tmp.alloc(&deg_s, n_rows);
    if (e == hipSuccess) e = tmp.alloc(&row_id, n_rows);
    if (e == hipSuccess) e = tmp.alloc(&order, n_rows);
    if (e == hipSuccess) e = tmp.alloc(&rowmap, n_rows);
    if (e == hipSuccess) e = tmp.alloc(&cnt, 4);
    own(&g->d_slot16, (size_t) n_rows8 + 8);
    if (e != hipSuccess) return bail("scratch");
    e = hipMemsetAsync(g->d_slot16, 0, sizeof(uint16_t) * ((size_t) n_rows8 + 8), s);// no row is a hub yet
    const uint32_t max_hub_ranges = std::min(32u, env_u32("SPLAHIP_COB_HUB_RANGES", 32));// (16 until late in round 3: R-MAT 23 has 22 ranges' worth of rows of degree >= 2 x average; alone neutral, and what lets the natural ranges grow there)
    // hub candidates: degree >= max(16, 2 x average); SPLAHIP_COB_HUB_MIN_DEGREE sets the threshold itself (tests: hub ranges on small inputs)
    const uint32_t thr            = env_u32("SPLAHIP_COB_HUB_MIN_DEGREE", std::max<uint32_t>(16, (uint32_t) (2ull * nnz / n_rows)));
    uint32_t       H              = 0;
    if (max_hub_ranges && e == hipSuccess) {
        hipLaunchKernelGGL(cob_deg_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, n_rows, d_Ap, deg, row_id);
        size_t bytes = 0;
        e            = rocprim::radix_sort_pairs_desc(nullptr, bytes, deg, deg_s, row_id, order, (size_t) n_rows, 0, 32, s);
        void* st_tmp = nullptr;
        if (e == hipSuccess) e = device_malloc(&st_tmp, bytes ? bytes : 4);
        if (e == hipSuccess) {
            tmp.ptrs.push_back(st_tmp);
            e = rocprim::radix_sort_pairs_desc(st_tmp, bytes, deg, deg_s, row_id, order, (size_t) n_rows, 0, 32, s);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cob_count_ge_kernel, dim3(1), dim3(1), 0, s, n_rows, deg_s, thr, cnt);
            e = hipMemcpyAsync(&H, cnt, 4, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("hub rows");
        H = (uint32_t) std::min<uint64_t>(H, (uint64_t) max_hub_ranges * R);
        if (H < R / 8) H = 0;// not a skewed matrix: every row stays in its natural range
        if (Rn > R) {// the longest row left to the natural ranges: the first one behind the hub rows in degree order
            uint32_t longest = 0;
            if (H < n_rows) e = hipMemcpy(&longest, deg_s + H, 4, hipMemcpyDeviceToHost);
            if (e != hipSuccess) return bail("hub rows");
            if (longest > env_u32("SPLAHIP_COB_NAT_MAX_DEGREE", 64)) Rn = R;
        }
        if (Rn > R && nat_auto) {
            // ... and only where a natural range of that size still is ONE piece of the launch: a range dearer than a workgroup slot's share is cut, and then
            // every piece of it merges 128 KB into the shared accumulator and the last one reads them back — R-MAT 22: 161 us against 145 with the ranges
            // left at R, where R-MAT 23 / 24 (larger shares) gain: 297 against 304 / 640 against 700 us (profiles/r03_notes.md 8).  Estimate before anything is
            // built: a natural entry costs ~1.7 hub entries (its gather line is mostly its own), a range holds Rn rows of the natural rows' average degree.
            std::vector<uint32_t> hd(H);
            uint32_t              n_nonempty = 0;
            if (H) e = hipMemcpy(hd.data(), deg_s, sizeof(uint32_t) * H, hipMemcpyDeviceToHost);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(cob_count_ge_kernel, dim3(1), dim3(1), 0, s, n_rows, deg_s, 1u, cnt);
                e = hipMemcpyAsync(&n_nonempty, cnt, 4, hipMemcpyDeviceToHost, s);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return bail("hub rows");
            uint64_t hub_e = 0;
            for (uint32_t d : hd) hub_e += d;
            const double nat_e = (double) nnz - (double) hub_e, nat_rows = (double) n_nonempty - (double) H;
            const double share = ((double) hub_e + 1.7 * nat_e) / std::max(1u, compute_units());
            if (nat_rows <= 0 || 1.7 * nat_e / nat_rows * Rn > 0.8 * share) Rn = R;
        }
    } else
        Rn = R;
    // (larger natural ranges keep an fp32 sum in float words, added in no fixed order: only where every natural row is short — <= 64 entries: a rounding error
    // of at most 64 x 2^-24 of the sum of the magnitudes, next to the 1e-5 bar; the reference's own sequential fold is no closer — do the ranges grow.
    // A matrix without hub rows, or with long rows left over behind 16 hub ranges, keeps ranges of R slots and double slices everywhere.)
    g->Rn          = Rn;
    g->acc_stride  = std::max(R, Rn);
    g->half_groups = Rn == R && 2 * ((size_t) R + kCobLdsExtra) * 8 <= 160 * 1024 && env_u32("SPLAHIP_COB_HALF_GROUPS", 1) != 0;
    if (H && env_u32("SPLAHIP_COB_HUB_ROW_ORDER", 1)) {
        unsigned long long *key = nullptr, *key_s = nullptr;
        e                       = tmp.alloc(&key, H);
        if (e == hipSuccess) e = tmp.alloc(&key_s, H);
        size_t bytes = 0;
        if (e == hipSuccess) e = rocprim::radix_sort_keys(nullptr, bytes, key, key_s, (size_t) H, 0, 36, s);
        void* st_tmp = nullptr;
        if (e == hipSuccess) e = device_malloc(&st_tmp, bytes ? bytes : 4);
        if (e == hipSuccess) {
            tmp.ptrs.push_back(st_tmp);
            hipLaunchKernelGGL(cob_hub_key_kernel, dim3((H + 255) / 256), dim3(256), 0, s, H, R, order, key);
            e = rocprim::radix_sort_keys(st_tmp, bytes, key, key_s, (size_t) H, 0, 36, s);
            hipLaunchKernelGGL(cob_hub_unkey_kernel, dim3((H + 255) / 256), dim3(256), 0, s, H, key_s, order);
        }
        if (e != hipSuccess) return bail("hub row order");
    }
    const uint32_t n_hub_ranges = (H + R - 1) / R;
    g->n_hub_ranges             = n_hub_ranges;
    g->heavy_slots              = std::min(H, 2 * R);// the mask probe's exact stratum: the rows of the first two hub ranges ...
    g->heavy_deg                = 0xFFFFFFFFu;
    if (g->heavy_slots) {// ... = the rows of at least this degree
        e = hipMemcpy(&g->heavy_deg, deg_s + g->heavy_slots - 1, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return bail("hub rows");
    }
    g->n_hub_rows               = H;
    own(&g->d_hub_row, (size_t) n_hub_ranges * R);
    if (e != hipSuccess) return bail("hub list");
    if (H) hipLaunchKernelGGL(cob_hub_kernel, dim3((n_hub_ranges * R + 255) / 256), dim3(256), 0, s, H, n_hub_ranges * R, R, order, rowmap, g->d_slot16, g->d_hub_row);
    // natural ranges: one greedy pass on the host over the per-8-row counts of slot-taking rows; a range closes when the next group would overflow
    // its R slots, or its span a bound that keeps one workgroup's row pass short
    std::vector<uint8_t>  h_cnt8(n_rows8 / 8);
    std::vector<uint32_t> h_gmap(n_rows8 / 8), nat_row0;
    {
        uint8_t* cnt8 = nullptr;
        e             = tmp.alloc(&cnt8, h_cnt8.size());
        if (e != hipSuccess) return bail("scratch");
        hipLaunchKernelGGL(cob_group8_kernel, dim3((n_rows8 / 8 + 255) / 256), dim3(256), 0, s, n_rows, d_Ap, g->d_slot16, cnt8);
        e = hipMemcpyAsync(h_cnt8.data(), cnt8, h_cnt8.size(), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("row counts");
        const uint32_t span_cap = std::max(8u, env_u32("SPLAHIP_COB_SPAN_CAP", 8) * Rn);
        uint32_t       q = n_hub_ranges, used = 0, span = 0;
        nat_row0.push_back(0);
        for (size_t gi = 0; gi < h_cnt8.size(); ++gi) {
            if (used + h_cnt8[gi] > Rn || span + 8 > span_cap) {
                nat_row0.push_back((uint32_t) gi * 8);
                ++q, used = 0, span = 0;
            }
            h_gmap[gi] = (q << kSlotBits) | used;
            used += h_cnt8[gi];
            span += 8;
        }
        nat_row0.push_back(n_rows);
    }
    const uint32_t n_nat    = (uint32_t) nat_row0.size() - 1;
    const uint32_t n_ranges = n_hub_ranges + n_nat;
    g->n_ranges             = n_ranges;
    const uint64_t n_groups64 = (uint64_t) n_ranges << win_bits;
    // every (range, window) group pads its last chunk: with more groups than nnz / 512 the copy would be mostly padding (a hypersparse
    // matrix) — the CSR kernels keep such a matrix
    if (n_groups64 > (uint64_t) nnz / 512 + 64 || n_groups64 >= (1u << 30) || n_ranges >= (1u << 16)) {
        cob_release(g);
        return SPLAHIP_OK;
    }
    const uint32_t n_groups = (uint32_t) n_groups64;
    SPLAHIP_REQUIRE(kSlotBits + cbits + bits_for(n_ranges) <= 64, "key does not fit 64 bits");
    {
        uint32_t* gmap = nullptr;
        e              = tmp.alloc(&gmap, h_gmap.size());
        if (e == hipSuccess) e = hipMemcpyAsync(gmap, h_gmap.data(), 4 * h_gmap.size(), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return bail("row map");
        hipLaunchKernelGGL(cob_rowmap_kernel, dim3((n_rows8 / 8 + 255) / 256), dim3(256), 0, s, n_rows, n_rows8, d_Ap, gmap, rowmap, g->d_slot16);
        e = hipStreamSynchronize(s);// h_gmap is read by the copy
        if (e != hipSuccess) return bail("row map");
    }

    stamp("hub rows, ranges, row map");
    // 1 + 2: keys, stable sort on the (range, column) bits
    unsigned long long *key = nullptr, *key_s = nullptr;
    uint32_t*           val_s = nullptr;
    e                         = tmp.alloc(&key, nnz);
    if (e == hipSuccess) e = tmp.alloc(&key_s, nnz);
    if (e == hipSuccess) e = tmp.alloc(&val_s, nnz);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_key_kernel, dim3((nnz + kKeyTile - 1) / kKeyTile), dim3(256), 0, s, n_rows, nnz, d_Ap, d_Aj, rowmap, cbits, key, col_map, g->n_hot_pad);
    e = hipGetLastError();
    {
        size_t      bytes = 0;
        const auto* vin   = static_cast<const uint32_t*>(d_Ax);
        const unsigned b0 = kSlotBits, b1 = kSlotBits + cbits + bits_for(n_ranges);
        if (b1 == b0) {// one range, one column: the entries are in order already
            if (e == hipSuccess) e = hipMemcpyAsync(key_s, key, sizeof(unsigned long long) * nnz, hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(val_s, vin, sizeof(uint32_t) * nnz, hipMemcpyDeviceToDevice, s);
        } else {
            if (e == hipSuccess) e = rocprim::radix_sort_pairs(nullptr, bytes, key, key_s, vin, val_s, (size_t) nnz, b0, b1, s);
            void* st_tmp = nullptr;
            if (e == hipSuccess) e = device_malloc(&st_tmp, bytes ? bytes : 4);
            if (e == hipSuccess) {
                tmp.ptrs.push_back(st_tmp);
                e = rocprim::radix_sort_pairs(st_tmp, bytes, key, key_s, vin, val_s, (size_t) nnz, b0, b1, s);
            }
        }
    }
    if (e != hipSuccess) return bail("sort");

    stamp("keys + radix sort");
    // 3: groups -> chunks
    uint32_t *start = nullptr, *chunk_first = nullptr;
    e = tmp.alloc(&start, (size_t) n_groups + 1);
    if (e == hipSuccess) e = tmp.alloc(&chunk_first, (size_t) n_groups + 1);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_group_kernel, dim3((n_groups + 256) / 256), dim3(256), 0, s, nnz, key_s, kSlotBits + col_bits, n_groups, start);
    // slots per entry (itself + the bridging slots of a column gap of more than 255), their running sum: where an entry's slots start
    uint32_t *wgt = nullptr, *pos = nullptr;
    e = tmp.alloc(&wgt, (size_t) nnz + 1);
    if (e == hipSuccess) e = tmp.alloc(&pos, (size_t) nnz + 1);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_weight_kernel, dim3((nnz + 256) / 256), dim3(256), 0, s, nnz, key_s, col_bits, cbits, wgt);
    {
        size_t bytes = 0;
        e            = rocprim::exclusive_scan(nullptr, bytes, wgt, pos, 0u, (size_t) nnz + 1, rocprim::plus<uint32_t>(), s);
        void* st_tmp = nullptr;
        if (e == hipSuccess) e = device_malloc(&st_tmp, bytes ? bytes : 4);
        if (e == hipSuccess) {
            tmp.ptrs.push_back(st_tmp);
            e = rocprim::exclusive_scan(st_tmp, bytes, wgt, pos, 0u, (size_t) nnz + 1, rocprim::plus<uint32_t>(), s);
        }
        if (e != hipSuccess) return bail("slot scan");
    }
    hipLaunchKernelGGL(cob_chunk_count_kernel, dim3((n_groups + 255) / 256), dim3(256), 0, s, n_groups, start, pos, chunk_first);
    hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, chunk_first, n_groups);
    e = hipGetLastError();
    uint32_t n_chunks = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&n_chunks, chunk_first + n_groups, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail("groups");
    g->n_chunks = n_chunks;

    stamp("groups -> chunks");
    // 4: the packed streams
    own(&g->d_sl, ((size_t) n_chunks + 2) * kCobChunk);// + two all-padding chunks (the main loop's steps beyond a piece's end walk one: hub / natural slot values)
    own(&g->d_dl, ((size_t) n_chunks + 2) * kCobChunk);
    own(&g->d_ax, ((size_t) n_chunks + 2) * kCobChunk);
    own(&g->d_hdr, (size_t) n_chunks + 2);
    if (e != hipSuccess) return bail("hipMalloc");
    hipLaunchKernelGGL(cob_padchunk_kernel, dim3(1), dim3(256), 0, s, n_chunks, R, g->d_sl, g->d_dl, g->d_ax, g->d_hdr);
    hipLaunchKernelGGL(cob_padchunk_kernel, dim3(1), dim3(256), 0, s, n_chunks + 1, Rn, g->d_sl, g->d_dl, g->d_ax, g->d_hdr);
    uint32_t* chunk_lines = nullptr;
    e                     = tmp.alloc(&chunk_lines, (size_t) n_chunks + 1);
    if (e != hipSuccess) return bail("scratch");
    if (n_chunks)
        hipLaunchKernelGGL(cob_scatter_kernel, dim3(n_chunks), dim3(256), 0, s, key_s, val_s, start, pos, chunk_first, n_groups, R, Rn, n_hub_ranges, cbits, win_bits, g->d_sl, g->d_dl,
                           g->d_ax, g->d_hdr, chunk_lines);
    e = hipGetLastError();
    if (e != hipSuccess) return bail("scatter");
    uint32_t* negzero_word = nullptr;
    e                      = tmp.alloc(&negzero_word, 1);
    if (e == hipSuccess) e = hipMemsetAsync(negzero_word, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return bail("scratch");
    hipLaunchKernelGGL(cob_negzero_kernel, dim3(1024), dim3(256), 0, s, (size_t) n_chunks * kCobChunk, g->d_ax, negzero_word);

    stamp("packed streams (scatter)");
    // 5: the work schedules, on the host (a few hundred pieces).  A piece = consecutive chunks of ONE range (the LDS slice belongs to a range) =
    // one workgroup.  What a chunk costs a CU is its memory requests: 16 lines of the two streams + the distinct lines of v its gathers touch
    // (round 3 timeline, profiles/r03_notes.md: a hub piece walks a chunk in 80 - 100 ns, a cold natural piece in 150 - 700 ns — a CU's miss queue
    // holds ~76 requests whatever they bring), so pieces are cut by that cost, not by chunk count: a range that costs at most `share` — what one
    // workgroup slot of the chip would get if the cost were dealt evenly, times SPLAHIP_COB_SHARE_PCT % — is a single piece and finishes its rows
    // itself; a dearer range is cut into pieces of equal cost that meet in the range's shared accumulator (mxv_cob.hpp).  Workgroups are launched
    // dearest piece first, so the cheap pieces fill the tail.  tile_chunks > 0 caps a piece's chunks explicitly (tests: many pieces on small inputs).
    std::vector<uint32_t> h_first((size_t) n_groups + 1), h_lines((size_t) n_chunks + 1);
    e = hipMemcpyAsync(h_first.data(), chunk_first, sizeof(uint32_t) * h_first.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && n_chunks) e = hipMemcpyAsync(h_lines.data(), chunk_lines, sizeof(uint32_t) * n_chunks, hipMemcpyDeviceToHost, s);
    uint32_t h_negzero = 1;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_negzero, negzero_word, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail("schedule");
    g->ax_negzero = h_negzero;
    std::vector<uint32_t> range_chunk((size_t) n_ranges + 1);
    for (uint32_t q = 0; q <= n_ranges; ++q) range_chunk[q] = h_first[(size_t) q << win_bits];
    // cost of a chunk in units of one stream line's stay in the miss queue: 14 stream lines + its gather lines weighted SPLAHIP_COB_LINE_PCT % (default 40: measured, a cold chunk of ~230 lines walks in ~340 ns, a hub chunk in ~85;
    // part of them hit the L2); cost_pre[c] = cost of the chunks before c
    const uint32_t        line_pct = env_u32("SPLAHIP_COB_LINE_PCT", 40);// (25 / 40 / 60 on one box, twice: R-MAT 22 151.5 / 145.2 / 147.9 us, R-MAT 23 307.5 / 304.1 / 308.7)
    std::vector<uint64_t> cost_pre((size_t) n_chunks + 1, 0);
    for (uint32_t c = 0; c < n_chunks; ++c) cost_pre[c + 1] = cost_pre[c] + 1400 + (uint64_t) h_lines[c] * line_pct;// (a chunk's three streams: 14 lines)
    own(&g->d_probe, 8);
    if (e == hipSuccess) e = hipMemsetAsync(g->d_probe, 0, sizeof(uint32_t) * 8, s);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(const_cast<uint32_t**>(&g->h_probe)), 4 * sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) {
        g->h_probe[0] = g->h_probe[1] = g->h_probe[2] = g->h_probe[3] = 0;
        e = hipHostGetDevicePointer(reinterpret_cast<void**>(&g->d_h_probe), const_cast<uint32_t*>(g->h_probe), 0);
    }
    if (e != hipSuccess) return bail("probe words");
    const uint32_t cus = compute_units();
    const size_t   lds_limit = lds_optin_bytes();
    uint32_t       max_multi = 0;
    for (int w = 0; w < 2 && e == hipSuccess; ++w) {
        splahip_cob_sched_s& sc     = g->sched[w];
        // LDS of a workgroup: the larger of a hub slice and a natural slice.  8-byte schedule (fp32 PLUS): hub slices hold doubles; natural slices too when
        // they are no larger than the hub ones, floats otherwise (mxv_cob.hpp: COB_FADD)
        const size_t lds = w ? std::max(((size_t) R + kCobLdsExtra) * 8, ((size_t) Rn + kCobLdsExtra) * (Rn > R ? 4 : 8)) : ((size_t) Rn + kCobLdsExtra) * 4;
        sc.lds_bytes     = (uint32_t) lds;
        // ONE 16-wave workgroup per CU, also where two 4-byte slices would fit together: measured on int32 min-plus (value form), one / two per CU: R-MAT 22
        // 148 / 185 us, R-MAT 23 264 / 283, R-MAT 24 520 / 546 — two workgroups compete for the same ~76 miss-queue slots of the CU, each with half the
        // waves' worth of requests in flight per piece, and the launch has twice as many piece starts and merges.  SPLAHIP_COB_PER_CU=2 for A/B runs.
        const uint32_t       per_cu = (uint32_t) std::min<size_t>(env_u32("SPLAHIP_COB_PER_CU", 1), (160 * 1024) / lds);// 1024-thread workgroups: at most 2 per CU
        if (per_cu == 0) continue;                                                         // this slice width does not fit: not served
        if (per_cu == 1 && 2 * lds <= 160 * 1024) sc.lds_bytes = 160 * 1024 / 2 + 1024;       // (one per CU asked for: ask for LDS a second workgroup cannot share)
        if (sc.lds_bytes > lds_limit) continue;                                            // a device with less LDS than gfx950: not served (n_tiles stays 0)
        const uint32_t slots = cus * per_cu;
        // piece size in % of a slot's even share of the cost.  Smaller pieces balance the tail better, larger ones mean fewer partial slices to merge,
        // and with pieces of similar cost the COUNT matters: 2.2 x slots pieces run in three rounds.  So the candidates are tried on a model of the launch
        // — dearest piece first onto the slot that frees first; a piece costs its chunks + ~4 us of zeroing and pipeline fill (+ ~3.5 us of merging
        // for a range in several pieces), a range's row pass goes to one of its pieces — and the share with the shortest modelled makespan is kept.
        // (cost unit: 1400 = one chunk's streams, 14 lines ~ 83 ns.)  SPLAHIP_COB_SHARE_PCT fixes the share instead.
        std::vector<uint32_t> tiles, ranges((size_t) n_ranges * 4);
        std::vector<uint64_t> cost;
        uint32_t              n_multi = 0;
        // levels > 1: TAPERED pieces — of a range's cost the first 4/7 go into pieces of size a, the next 2/7 into pieces of a / 2, the last 1/7 into pieces of a / 4
        // (levels = 3; 2/3 + 1/3 for levels = 2): launched dearest first, the rounds end at a, 1.5 a, 1.75 a and the tail is made of small pieces
        const uint32_t taper_mode = env_u32("SPLAHIP_COB_TAPER", 0);
        const uint32_t round_chunks = env_u32("SPLAHIP_COB_ROUND_CHUNKS", 64);// chunks of one round of the kernel's loop: 16 waves x 4 steps
        auto cut = [&](uint32_t share_pct, uint32_t levels) {
            const uint64_t share = std::max<uint64_t>(1, (cost_pre[n_chunks] + slots - 1) / slots * share_pct / 100);
            tiles.clear(), cost.clear();
            n_multi = 0;
            for (uint32_t q = 0; q < n_ranges; ++q) {
                const uint32_t c0 = range_chunk[q], c1 = range_chunk[q + 1], cc = c1 - c0;
                const bool     hub = q < n_hub_ranges;
                const uint32_t row0 = hub ? 0u : nat_row0[q - n_hub_ranges], row1 = hub ? 0u : nat_row0[q - n_hub_ranges + 1];
                const uint64_t rc   = cost_pre[c1] - cost_pre[c0];
                // the cost marks at which the range is cut
                std::vector<uint64_t> marks;
                if (rc > share && cc > 1) {
                    const uint32_t L     = std::max(1u, std::min(levels, 5u));
                    const uint64_t denom = (1u << L) - 1u;// 1, 3, 7, 15, 31
                    uint64_t       at    = 0;
                    for (uint32_t l = 0; l < L; ++l) {
                        // taper 0: equal piece COUNTS per level (4/7, 2/7, 1/7 of the cost); 1 (SPLAHIP_COB_TAPER): equal COST per level — 1, 2, 4 ... times as many
                        // pieces from level to level, i.e. far more small pieces for the second half of the launch to balance with
                        const uint64_t part = taper_mode ? rc / L : rc * (1u << (L - 1 - l)) / denom, size = std::max<uint64_t>(1, share >> l);
                        const uint64_t np   = std::max<uint64_t>(1, (part + size - 1) / size);
                        for (uint64_t t = 1; t <= np; ++t) marks.push_back(at + part * t / np);
                        at += part;
                    }
                    marks.back() = rc;
                } else
                    marks.push_back(rc);
                // the row pass of the range (its last piece to arrive): ~12 us of round trips for a hub range's scattered rows; ~6 us + 2 us per 2^15 rows
                const uint64_t row_pass = hub ? 200000 : 100000 + (uint64_t) (row1 - row0);
                uint32_t       f = c0, first_tile = (uint32_t) (tiles.size() / 4);
                for (size_t t = 0; t < marks.size() && (f < c1 || t == 0); ++t) {
                    uint32_t end = c1;
                    if (t + 1 < marks.size()) {
                        end = (uint32_t) (std::lower_bound(cost_pre.begin() + f, cost_pre.begin() + c1, cost_pre[c0] + marks[t]) - cost_pre.begin());
                        end = std::min(std::max(end, f + 1), c1);
                        // a workgroup walks its piece 16 chunks a step, in rounds of 4 steps: a piece of a whole number of rounds leaves no wave walking the
                        // padding chunk (a 233-chunk piece walks 256).  Pieces of >= 4 rounds are cut at the nearest whole round (SPLAHIP_COB_ROUND_CHUNKS=0: as cut)
                        if (round_chunks && end - f >= 4 * round_chunks && end < c1) {
                            const uint32_t len = (end - f + round_chunks / 2) / round_chunks * round_chunks;
                            if (f + len < c1) end = f + len;
                        }
                    }
                    if (tile_chunks) end = std::min(end, f + tile_chunks);
                    tiles.insert(tiles.end(), {f, end - f, q, kCobNoSlab});
                    cost.push_back(cost_pre[end] - cost_pre[f] + 67000);
                    f = end;
                }
                while (f < c1) {// (an explicit cap left chunks over: more pieces of at most tile_chunks)
                    const uint32_t end = tile_chunks ? std::min(c1, f + tile_chunks) : c1;
                    tiles.insert(tiles.end(), {f, end - f, q, kCobNoSlab});
                    cost.push_back(cost_pre[end] - cost_pre[f] + 67000);
                    f = end;
                }
                const uint32_t pieces = (uint32_t) (tiles.size() / 4) - first_tile;
                if (!(hub && g->hub_finish)) cost.back() += row_pass;// (the range's cheapest piece is launched last: it is the likely last arriver)
                if (hub && g->hub_finish) {// every piece merges, none runs the row pass (cob_hub_finish_kernel): accumulator q belongs to hub range q
                    for (uint32_t t = first_tile; t < first_tile + pieces; ++t) tiles[4 * (size_t) t + 3] = n_multi, cost[t] += 59000;
                    ++n_multi;
                    ranges[4 * (size_t) q + 0] = 0u, ranges[4 * (size_t) q + 1] = 0u;
                    ranges[4 * (size_t) q + 2] = q * R, ranges[4 * (size_t) q + 3] = kCobNoLast;
                    continue;
                }
                if (pieces > 1) {
                    for (uint32_t t = first_tile; t < first_tile + pieces; ++t) tiles[4 * (size_t) t + 3] = n_multi, cost[t] += 59000;
                    ++n_multi;
                }
                ranges[4 * (size_t) q + 0] = row0, ranges[4 * (size_t) q + 1] = row1, ranges[4 * (size_t) q + 2] = hub ? q * R : 0xFFFFFFFFu, ranges[4 * (size_t) q + 3] = pieces;
            }
        };
        // launch order.  0: dearest piece first.  1 (SPLAHIP_COB_ORDER): RANGE-MAJOR — the natural ranges' pieces first (latency-bound walks and long row
        // passes: all under the hub pieces' streams), then the hub ranges one after the other, dearest range first, each range's pieces dearest first — so a
        // range's row pass (its last arriver's ~12 - 20 us) falls under the next ranges' streams and only the last range's is in the tail
        const uint32_t order_mode = env_u32("SPLAHIP_COB_ORDER", 2);// (2 since the last A/B of round 3: R-MAT 23 288 against 294 us, R-MAT 22 / 24 level)
        auto launch_order = [&]() {
            const uint32_t        nt = (uint32_t) (tiles.size() / 4);
            std::vector<uint32_t> o(nt);
            for (uint32_t t = 0; t < nt; ++t) o[t] = t;
            if (order_mode == 0) {
                std::stable_sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
                return o;
            }
            if (order_mode == 2) {// NATURAL pieces first (dearest first), then the hub pieces dearest first: a natural range's last piece ends with a read-back and a
                                  // row pass of ~20 us on its one CU and walks cold lines at 250 ns per chunk — launched last for being small, such pieces WERE the tail
                                  // of the launch (R-MAT 23: the last 55 us); a hub piece ends with 2.7 us of merging and is the better tail filler
                std::stable_sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) {
                    const bool hx = tiles[4 * (size_t) x + 2] < n_hub_ranges, hy = tiles[4 * (size_t) y + 2] < n_hub_ranges;
                    if (hx != hy) return !hx;
                    return cost[x] > cost[y];
                });
                return o;
            }
            std::vector<uint64_t> range_cost(n_ranges, 0);
            for (uint32_t t = 0; t < nt; ++t) range_cost[tiles[4 * (size_t) t + 2]] += cost[t];
            std::stable_sort(o.begin(), o.end(), [&](uint32_t x, uint32_t y) {
                const uint32_t qx = tiles[4 * (size_t) x + 2], qy = tiles[4 * (size_t) y + 2];
                const bool     hx = qx < n_hub_ranges, hy = qy < n_hub_ranges;
                if (hx != hy) return !hx;// natural pieces first
                if (!hx) return cost[x] > cost[y];
                if (qx != qy) return range_cost[qx] != range_cost[qy] ? range_cost[qx] > range_cost[qy] : qx < qy;
                return cost[x] > cost[y];
            });
            return o;
        };
        auto makespan = [&]() -> uint64_t {
            const std::vector<uint32_t> o = launch_order();
            std::priority_queue<uint64_t, std::vector<uint64_t>, std::greater<uint64_t>> free_at;
            for (uint32_t k = 0; k < slots; ++k) free_at.push(0);
            uint64_t end = 0;
            for (uint32_t t : o) {
                const uint64_t e2 = free_at.top() + cost[t];
                free_at.pop();
                free_at.push(e2);
                end = std::max(end, e2);
            }
            return end;
        };
        // (three size levels measured best at both bench scales whatever the model preferred — R-MAT 22: 154 us against 166 / 172 with two / one level, R-MAT 23: 308
        // against 336 / 327 — so the level count is fixed and the model picks the share; SPLAHIP_COB_LEVELS=0 lets it pick both)
        // (round 5: a FOURTH level for the schedule of the 4-byte slices where a slot has >= 3500 chunks to walk — the min-plus step at R-MAT 24: the smallest pieces are
        // then still ~5 rounds of the loop.  R-MAT 24 min-plus 544 - 548 -> 522 - 529 us; the fp32 sums (double slices) gain nothing there (609 - 613 -> 604 - 613 us, the
        // PageRank step level) and keep three; with fewer chunks per slot four levels lose: R-MAT 22 138 -> 162 us, R-MAT 23 271 -> 293 at the model's share —
        // tools/gpu_r5_levels.sh)
        const uint32_t levels_default = w == 0 && n_chunks / std::max(1u, slots) >= 3500u ? 4u : 3u;
        uint32_t       share_pct = env_u32("SPLAHIP_COB_SHARE_PCT", 0), levels = env_u32("SPLAHIP_COB_LEVELS", levels_default);
        if (share_pct == 0 || levels == 0) {
            uint64_t       best = ~0ull;
            const uint32_t pct_fixed = share_pct, lev_fixed = levels;
            for (uint32_t lv : {1u, 2u, 3u, 4u}) {
                if (lev_fixed ? lv != lev_fixed : lv == 4u) continue;// (LEVELS=0: the model picks among 1 - 3)
                for (uint32_t pct : {140u, 120u, 110u, 100u, 90u, 80u, 70u, 65u, 60u, 55u, 50u, 45u, 40u, 35u, 30u, 25u}) {
                    if (pct_fixed && pct != pct_fixed) continue;
                    cut(pct, lv);
                    const uint64_t m = makespan();
                    if (m < best) best = m, share_pct = pct, levels = lv;
                }
            }
            if (best == ~0ull) share_pct = pct_fixed ? pct_fixed : 60, levels = lev_fixed ? lev_fixed : 1;
        }
        cut(share_pct, levels);
        const uint32_t              n_tiles = (uint32_t) (tiles.size() / 4);
        const std::vector<uint32_t> order_t = launch_order();
        std::vector<uint32_t>       sorted(tiles.size());
        for (uint32_t t = 0; t < n_tiles; ++t)
            for (int k = 0; k < 4; ++k) sorted[4 * t + k] = tiles[4 * order_t[t] + k];
        if (knob_env("SPLAHIP_COB_TRACE"))
            std::fprintf(stderr, "[cob] schedule %d: R %u / %u, %u hub rows in %u hub ranges + %u natural ranges; slots %u, cost %.1f M (%.1f %% gathers), piece <= %u %% of a share in %u size levels, %u pieces, "
                                 "%u ranges in several pieces, modelled makespan %.1f us (even: %.1f)\n",
                         w, R, Rn, H, n_hub_ranges, n_nat, slots, cost_pre[n_chunks] / 1e8, 100.0 - 100.0 * 1400.0 * n_chunks / std::max<uint64_t>(1, cost_pre[n_chunks]), share_pct, levels, n_tiles, n_multi,
                         makespan() * 83e-3 / 1400, cost_pre[n_chunks] * 83e-3 / 1400 / slots);
        sc.n_tiles = n_tiles;
        sc.n_multi = n_multi;
        max_multi  = std::max(max_multi, n_multi);
        own(&sc.d_tiles, sorted.size());
        own(&sc.d_ranges, ranges.size());
        if (e == hipSuccess && !sorted.empty()) e = hipMemcpyAsync(sc.d_tiles, sorted.data(), 4 * sorted.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(sc.d_ranges, ranges.data(), 4 * ranges.size(), hipMemcpyHostToDevice, s);
        // The run-time schedule (mxv_cob_dyn.hpp): `slots` persistent workgroups; the job list only says where they START.  A hub range is listed once per
        // worker its modelled cost is worth (cost / a slot's even share, rounded), a natural range once (a second worker would cost both a 128 KB merge);
        // order: as many natural ranges as the hub workers leave slots for, dearest first (a natural range is walked by ONE workgroup at cold-line speed:
        // the longest ones must not start late), then the hub ranges' workers, then the remaining natural ranges dearest first — the natural workers take
        // them as they finish.  Whatever the model got wrong is evened out at run time: a workgroup that finds the list used up joins the range with the
        // most unclaimed quads per worker.  tile_chunks (tests) / SPLAHIP_COB_DYN_WORKERS list every range several times: the merge path on small inputs.
        std::vector<uint32_t> jobs;
        {
            const uint32_t dyn_workers = env_u32("SPLAHIP_COB_DYN_WORKERS", 0);
            std::vector<std::pair<uint64_t, uint32_t>> nat;
            std::vector<uint64_t>                      hub_cost(n_hub_ranges, 0);
            uint64_t                                   hub_total = 0;
            for (uint32_t q = 0; q < n_ranges; ++q) {
                const uint32_t c0 = range_chunk[q], c1 = range_chunk[q + 1];
                const bool     hub = q < n_hub_ranges;
                const uint64_t rc  = cost_pre[c1] - cost_pre[c0];
                if (hub) hub_cost[q] = rc, hub_total += rc;
                else
                    nat.emplace_back(rc, q);
            }
            std::stable_sort(nat.begin(), nat.end(), [](const auto& x, const auto& y) { return x.first > y.first; });
            // a natural range is walked by ONE workgroup from start to end at cold-line speed (measured: ~200 ns per chunk against ~95 in a hub range — a
            // 1500-chunk range is as long as the whole launch), so the natural ranges start FIRST, as many as there are slots beside one worker per hub
            // range; the hub ranges share the slots that are left in proportion to their cost (they are elastic: whoever runs out of work joins them)
            const uint32_t first_nat  = (uint32_t) std::min<size_t>(nat.size(), slots > n_hub_ranges ? slots - n_hub_ranges : 0u);
            for (uint32_t i = 0; i < first_nat; ++i) jobs.push_back(nat[i].second);
            // ... and a natural range that ALONE would take most of the launch gets a second worker from the start (they then share its rounds and merge:
            // ~25 us of a CU, against a pole of up to 1.2 x the launch at R-MAT 23).  Lengths from measured rates: ~95 ns per hub chunk, ~210 per natural chunk.
            uint32_t n_second = 0;
            {
                const uint64_t hub_chunks = n_hub_ranges ? range_chunk[n_hub_ranges] : 0u, nat_chunks = n_chunks - hub_chunks;
                const uint64_t ideal_ns   = (hub_chunks * 95u + nat_chunks * 210u) / std::max(1u, slots);
                const uint32_t pct        = env_u32("SPLAHIP_COB_DYN_NAT_SOLO_PCT", 80);
                for (uint32_t i = 0; i < first_nat && first_nat + n_second + n_hub_ranges < slots; ++i) {
                    const uint32_t q = nat[i].second;
                    if ((uint64_t) (range_chunk[q + 1] - range_chunk[q]) * 210u * 100u > ideal_ns * pct) jobs.push_back(q), ++n_second;
                }
            }
            const uint32_t hub_slots = std::max(n_hub_ranges, slots > first_nat + n_second ? slots - first_nat - n_second : 0u);
            uint32_t given = 0;
            uint64_t run   = 0;
            for (uint32_t q = 0; q < n_hub_ranges; ++q) {// largest-remainder shares of hub_slots, at least one each
                run += hub_cost[q];
                uint32_t upto = hub_total ? (uint32_t) ((run * hub_slots + hub_total / 2) / hub_total) : q + 1;
                uint32_t k    = std::max(1u, upto > given ? upto - given : 0u);
                if (dyn_workers) k = dyn_workers;
                if (tile_chunks) k = std::max(k, std::min(8u, (range_chunk[q + 1] - range_chunk[q] + tile_chunks - 1) / tile_chunks));
                given += k;
                jobs.insert(jobs.end(), k, q);
            }
            for (size_t i = first_nat; i < nat.size(); ++i) jobs.push_back(nat[i].second);
            if (dyn_workers > 1 || tile_chunks)// (tests: the merge path of the natural ranges on small inputs)
                for (const auto& nq : nat) jobs.insert(jobs.end(), std::min(7u, std::max(dyn_workers, 2u) - 1u), nq.second);
        }
        sc.slots  = slots;
        sc.n_jobs = (uint32_t) jobs.size();
        own(&sc.d_jobs, jobs.size());
        own(&sc.d_dyn_ctl, 2);
        if (e == hipSuccess && !jobs.empty()) e = hipMemcpyAsync(sc.d_jobs, jobs.data(), 4 * jobs.size(), hipMemcpyHostToDevice, s);
        if (knob_env("SPLAHIP_COB_TRACE")) std::fprintf(stderr, "[cob] run-time schedule %d: %u slots, %zu listed jobs over %u ranges\n", w, slots, jobs.size(), n_ranges);
        if (e == hipSuccess) e = hipStreamSynchronize(s);// the vectors go out of scope
    }
    // SPLAHIP_COB_DYN=1: the run-time schedule of mxv_cob_dyn.hpp instead of the static pieces.  Measured (profiles/r05_notes.md 1): correct everywhere, and NOT
    // faster — R-MAT 23 fp32 plus-times 282 us against 270 for the main launch (300 against 284 per call), R-MAT 22 165 against 143: the launch still ends with
    // ~50 us in which few CUs walk the last rounds at latency speed, and every worker that joins a range late pays its zeroing, pipeline fill and merge.  Off by default.
    g->dyn         = env_u32("SPLAHIP_COB_DYN", 0) != 0;
    g->dyn_min_hub = env_u32("SPLAHIP_COB_DYN_MIN_HUB", 2400);// unclaimed weighted cost per worker: chunks x 1/16 plain-stream chunks (~5 ns each): 2400 ~ 12 us of walking (joining costs ~8 us; a natural range ~25)
    g->dyn_min_nat = env_u32("SPLAHIP_COB_DYN_MIN_NAT", 8000);
    {
        std::vector<uint32_t> spans((size_t) n_ranges * 2);
        for (uint32_t q = 0; q < n_ranges; ++q) spans[2 * (size_t) q] = range_chunk[q], spans[2 * (size_t) q + 1] = range_chunk[q + 1] - range_chunk[q];
        own(&g->d_spans, spans.size());
        own(&g->d_dyn_words, (size_t) n_ranges * 3);// cursors, states, finish flags
        if (e == hipSuccess) e = hipMemcpyAsync(g->d_spans, spans.data(), 4 * spans.size(), hipMemcpyHostToDevice, s);
        // modelled cost of one chunk of a range in 1/16 of a plain stream chunk's (the steal rule weighs a range's unclaimed chunks with it); the ranges
        // as the row pass wants them (a lone worker of a natural range finishes its rows itself)
        std::vector<uint16_t> weight(n_ranges);
        std::vector<uint32_t> dyn_ranges((size_t) n_ranges * 4);
        for (uint32_t q = 0; q < n_ranges; ++q) {
            const uint32_t c0 = range_chunk[q], c1 = range_chunk[q + 1];
            weight[q]         = (uint16_t) std::min<uint64_t>(65535, c1 > c0 ? (cost_pre[c1] - cost_pre[c0]) / (100ull * (c1 - c0)) : 16);
            const bool hub    = q < n_hub_ranges;
            dyn_ranges[4 * (size_t) q + 0] = hub ? 0u : nat_row0[q - n_hub_ranges], dyn_ranges[4 * (size_t) q + 1] = hub ? 0u : nat_row0[q - n_hub_ranges + 1];
            dyn_ranges[4 * (size_t) q + 2] = hub ? q * R : 0xFFFFFFFFu, dyn_ranges[4 * (size_t) q + 3] = 1u;
        }
        own(&g->d_dyn_weight, weight.size());
        own(&g->d_dyn_ranges, dyn_ranges.size());
        if (e == hipSuccess) e = hipMemcpyAsync(g->d_dyn_weight, weight.data(), 2 * weight.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(g->d_dyn_ranges, dyn_ranges.data(), 4 * dyn_ranges.size(), hipMemcpyHostToDevice, s);
        // the finishing kernel's row tasks: at most 2048 consecutive rows of one natural range each (row0, row1, -, range)
        std::vector<uint32_t> tasks;
        for (uint32_t q = n_hub_ranges; q < n_ranges; ++q)
            for (uint32_t r0 = nat_row0[q - n_hub_ranges]; r0 < nat_row0[q - n_hub_ranges + 1]; r0 += 2048u)
                tasks.insert(tasks.end(), {r0, std::min(r0 + 2048u, nat_row0[q - n_hub_ranges + 1]), 0xFFFFFFFFu, q});
        g->n_fin_tasks = (uint32_t) (tasks.size() / 4);
        own(&g->d_fin_tasks, tasks.size());
        if (e == hipSuccess && !tasks.empty()) e = hipMemcpyAsync(g->d_fin_tasks, tasks.data(), 4 * tasks.size(), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (g->dyn) max_multi = std::max(max_multi, n_ranges);// the run-time schedule (opt-in): any range may be shared, accumulator q belongs to range q (not sized for it otherwise)
    own(&g->d_acc, (size_t) max_multi * g->acc_stride);
    g->ticket_words = max_multi + 4 + 32;
    own(&g->d_ticket, (size_t) g->ticket_words);
    own(&g->d_vbits, ((size_t) g->n_vcols + 255) / 256 * 8 + 2);// every wave of cob_vbits_kernel writes its two words
    if (e != hipSuccess) return bail("schedule upload");
    stamp("schedules");
    {// one value word in the whole matrix?  (the sparse-vector form then skips the value stream)
        uint32_t* rng = nullptr;
        e             = tmp.alloc(&rng, 2);
        const uint32_t init_rng[2] = {0xFFFFFFFFu, 0u};
        uint32_t       h_rng[2]    = {0, 1};
        if (e == hipSuccess) e = hipMemcpyAsync(rng, init_rng, 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(cob_ax_range_kernel, dim3(1024), dim3(256), 0, s, nnz, static_cast<const uint32_t*>(d_Ax), rng);
            e = hipMemcpyAsync(h_rng, rng, 8, hipMemcpyDeviceToHost, s);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail("value range");
        g->x_uniform = h_rng[0] == h_rng[1];
        g->x_bits    = h_rng[0];
    }
    stamp("value range");
    plan->cob = g;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_info(splahip_plan plan, uint32_t* n_tiles, uint32_t* n_chunks, uint32_t* n_multi, uint32_t* slots_per_range) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_cob_s* g = plan->cob;
    if (n_tiles) *n_tiles = g ? std::max(g->sched[0].n_tiles, g->sched[1].n_tiles) : 0;
    if (n_chunks) *n_chunks = g ? g->n_chunks : 0;
    if (n_multi) *n_multi = g ? std::max(g->sched[0].n_multi, g->sched[1].n_multi) : 0;
    if (slots_per_range) *slots_per_range = g ? g->R : 0;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_layout(splahip_plan plan, uint32_t* n_ranges, uint32_t* n_hub_ranges, uint32_t* n_hub_rows, uint64_t* bytes) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_cob_s* g = plan->cob;
    if (n_ranges) *n_ranges = g ? g->n_ranges : 0;
    if (n_hub_ranges) *n_hub_ranges = g ? g->n_hub_ranges : 0;
    if (n_hub_rows) *n_hub_rows = g ? g->n_hub_rows : 0;
    if (bytes) *bytes = g ? (uint64_t) g->bytes : 0;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_probe(splahip_plan plan, uint64_t* selected_entries, int* served_by_rows) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    if (selected_entries) *selected_entries = 0;
    if (served_by_rows) *served_by_rows = 0;
    const splahip_cob_s* g = plan->cob;
    if (!g || !g->h_probe) return SPLAHIP_OK;
    SPLAHIP_TRY(hipDeviceSynchronize());
    if (g->h_probe[3] == 0) return SPLAHIP_OK;// no probe has run
    if (selected_entries) *selected_entries = ((uint64_t) g->h_probe[1] << 32) | g->h_probe[0];
    if (served_by_rows) *served_by_rows = g->last_rows;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_relabel(splahip_plan plan, uint32_t* hot_columns) {
    SPLAHIP_REQUIRE(plan != nullptr && hot_columns != nullptr, "null argument");
    *hot_columns = plan->cob ? plan->cob->n_hot : 0;
    return SPLAHIP_OK;
}

extern "C" int splahip_csr_plan_cob_slots(splahip_plan plan, uint32_t* hub_slots, uint32_t* natural_slots) {
    SPLAHIP_REQUIRE(plan != nullptr, "plan is null");
    const splahip_cob_s* g = plan->cob;
    if (hub_slots) *hub_slots = g ? g->R : 0;
    if (natural_slots) *natural_slots = g ? g->Rn : 0;
    return SPLAHIP_OK;
}

#ifdef SPLAHIP_COB_TIMING
// DIAGNOSTIC BUILD ONLY: copy out the per-workgroup stamps of the last cob launch (8 x 8192 64-bit words)
extern "C" int splahip_debug_cob_timing(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(splahip::g_cob_t), sizeof(unsigned long long) * 8 * 8192) == hipSuccess ? SPLAHIP_OK : SPLAHIP_ERROR;
}
// the run-time schedule appends one record per (workgroup, range) visit: start over (and clear the records) before the launch to be looked at
extern "C" int splahip_debug_cob_timing_reset(void) {
    unsigned int zero = 0;
    void*        p    = nullptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(splahip::g_cob_tn), &zero, sizeof(zero)) != hipSuccess) return SPLAHIP_ERROR;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(splahip::g_cob_t)) != hipSuccess) return SPLAHIP_ERROR;
    return hipMemset(p, 0, sizeof(unsigned long long) * 8 * 8192) == hipSuccess ? SPLAHIP_OK : SPLAHIP_ERROR;
}
#endif
