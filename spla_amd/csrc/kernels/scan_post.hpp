// scan_post.hpp — the scan pass of a count / scan / write compaction that also POSTS the total to the calling host thread's mailbox (common.hpp): the host that waits
// for the count (a traversal level's next frontier, a conversion's entry count) gets it from the one-workgroup scan kernel — before the write pass has run, and
// without the mailbox launch that used to follow it.  Everything later in the stream is ordered behind the write pass as before.  SPLAHIP_SCAN_POST=0: the plain
// scan and a mailbox launch behind the write pass.
#pragma once

#include "block_utils.hpp"
#include "common.hpp"

namespace splahip {

    // chunk_scan_kernel (block_utils.hpp) + the post: counts[n] and mailbox slot 0 receive the total; extra words (n_extra <= 15, e.g. a call's gate words) go to slots 1 ..
    template<unsigned NT>
    __global__ __launch_bounds__(NT) void chunk_scan_post_kernel(uint32_t* __restrict__ counts, uint32_t n, unsigned long long* __restrict__ mailbox, uint32_t mail_seq,
                                                                 const uint32_t* __restrict__ extra, uint32_t n_extra) {
        __shared__ uint32_t s_tmp[NT / kWave + 1];
        __shared__ uint32_t s_carry;
        if (threadIdx.x == 0) s_carry = 0;
        __syncthreads();
        for (uint32_t base = 0; base < n; base += NT) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t x = i < n ? counts[i] : 0;
            uint32_t       total;
            const uint32_t ex    = block_exclusive_sum<NT>(x, s_tmp, total);
            const uint32_t carry = s_carry;
            if (i < n) counts[i] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) s_carry = carry + total;
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            counts[n] = s_carry;
            mailbox_store(mailbox, 0, mail_seq, s_carry);
        }
        if (threadIdx.x >= 1 && threadIdx.x <= n_extra) mailbox_store(mailbox, threadIdx.x, mail_seq, extra[threadIdx.x - 1]);
    }

    struct ScanPost {
        MailboxTicket mt;
        bool          posted = false;
    };
    // enqueue the scan of counts[0 .. n) (total into counts[n]); with the post when the mailbox is there and the knob allows
    inline hipError_t scan_launch(uint32_t* counts, uint32_t n, hipStream_t s, ScanPost* sp, const uint32_t* extra = nullptr, uint32_t n_extra = 0) {
        const char* e = knob_env("SPLAHIP_SCAN_POST");
        sp->posted    = !(e && e[0] == '0') && mailbox_begin(&sp->mt) == hipSuccess;
        if (sp->posted) hipLaunchKernelGGL(chunk_scan_post_kernel<1024>, dim3(1), dim3(1024), 0, s, counts, n, sp->mt.dev, sp->mt.seq, extra, n_extra);
        else
            hipLaunchKernelGGL(chunk_scan_kernel<1024>, dim3(1), dim3(1024), 0, s, counts, n);
        return hipGetLastError();
    }
    // the total on the host: the posted word (call this AFTER enqueueing the write pass), or a mailbox launch over d_total
    inline hipError_t scan_total(uint32_t* h_total, const uint32_t* d_total, const ScanPost& sp, hipStream_t s) {
        return sp.posted ? mailbox_end(h_total, 1, sp.mt, s) : fetch_words(h_total, d_total, 1, s);
    }

}// namespace splahip
