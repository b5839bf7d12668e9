// The 1-D row-block decomposition driven from C++ through the kernel-level C ABI alone (include/spla_hip.h: splahip_dist_* / splahip_shard_* / splahip_dist_sx_*)
// — what a host without Python binds.  One GPU: the matrix is cut into WORLD simulated ranks that are driven one after the other; each
// rank's shard computes its local-column part from its own slice of v, the "exchange" hands it the gathered vector (here: the dense all-gather through an
// RCCL communicator of world size 1 for WORLD == 1, a device copy otherwise; and the sparse exchange's pack / scatter kernels for a frontier-like v), and
// the remote-column part folds onto r.  Stitched rows must equal splahip_mxv_masked on the whole matrix bit for bit (int32 min-plus and plus-times).
// Built by oracle/Makefile (target usertests) against include/ + libspla_amd.so; run by tests/test_user_ops_gpu.py.
#include <spla_hip.h>
#include <spla_hip_ops.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(cond)                                                                                                     \
    do {                                                                                                                \
        if (!(cond)) {                                                                                                  \
            std::fprintf(stderr, "CHECK failed: %s (%s:%d) [%s]\n", #cond, __FILE__, __LINE__, splahip_last_error()); \
            return 1;                                                                                                   \
        }                                                                                                               \
    } while (0)

template<typename T>
static T* upload(const std::vector<T>& h) {
    void* d = nullptr;
    if (splahip_malloc(&d, sizeof(T) * std::max<size_t>(h.size(), 1)) != SPLAHIP_OK) return nullptr;
    if (!h.empty() && splahip_memcpy_h2d(d, h.data(), sizeof(T) * h.size(), nullptr) != SPLAHIP_OK) return nullptr;
    return static_cast<T*>(d);
}

int main() {
    if (splahip_init(0) != SPLAHIP_OK) {
        std::fprintf(stderr, "no usable device: %s\n", splahip_last_error());
        return 2;
    }
    // a skewed random matrix n x n (R-MAT-like: row and column drawn with a bias towards low ids), sorted, duplicate-free rows
    const uint32_t n = 20000;
    unsigned       seed = 2463534242u;
    auto           rnd  = [&]() { return seed ^= seed << 13, seed ^= seed >> 17, seed ^= seed << 5; };
    std::vector<std::vector<uint32_t>> rows(n);
    for (uint32_t e = 0; e < 16 * n; ++e) {
        uint32_t i = rnd() % n, j = rnd() % n;
        if (rnd() & 1) i = i % (n / 16 + 1);
        if (rnd() & 1) j = j % (n / 16 + 1);
        rows[i].push_back(j);
    }
    std::vector<uint32_t> Ap(n + 1, 0), Aj;
    std::vector<int32_t>  Ax;
    for (uint32_t i = 0; i < n; ++i) {
        std::sort(rows[i].begin(), rows[i].end());
        rows[i].erase(std::unique(rows[i].begin(), rows[i].end()), rows[i].end());
        for (uint32_t j : rows[i]) Aj.push_back(j), Ax.push_back(1 + (int32_t) (rnd() % 255));
        Ap[i + 1] = (uint32_t) Aj.size();
    }
    const uint32_t       nnz = Ap[n];
    const int32_t        INF = 1 << 30;
    std::vector<int32_t> v(n, INF), mask(n, 0), vsp(n, 0);
    for (uint32_t j = 0; j < n; ++j) {
        if (rnd() % 8 == 0) v[j] = (int32_t) (rnd() % 5000);
        if (rnd() % 3 == 0) mask[j] = 1;
        if (rnd() % 40 == 0) vsp[j] = 1;// a frontier-like vector for the sparse exchange
    }
    // the whole matrix, one device: the result every decomposition must reproduce
    uint32_t *   dAp = upload(Ap), *dAj = upload(Aj);
    int32_t *    dAx = upload(Ax), *dv = upload(v), *dmask = upload(mask), *dvsp = upload(vsp);
    CHECK(dAp && dAj && dAx && dv && dmask && dvsp);
    splahip_plan plan = nullptr;
    CHECK(splahip_csr_plan_create(&plan, n, nnz, dAp, nullptr) == SPLAHIP_OK);
    struct Case {
        int      mul, add, sel;
        int32_t  init;
        int32_t* vec;
        const std::vector<int32_t>* hvec;
    } cases[] = {{SPLAHIP_PLUS, SPLAHIP_MIN, SPLAHIP_EQZERO, INF, dv, &v}, {SPLAHIP_MULT, SPLAHIP_PLUS, SPLAHIP_ALWAYS, 0, dvsp, &vsp}};
    for (const Case& c : cases) {
        std::vector<int32_t> want(n);
        int32_t*             dr = upload(want);
        CHECK(splahip_mxv_masked(plan, SPLAHIP_INT, n, n, dAp, dAj, dAx, dmask, c.vec, dr, c.mul, c.add, c.sel, (uint32_t) c.init, 0, nullptr) == SPLAHIP_OK);
        CHECK(splahip_memcpy_d2h(want.data(), dr, 4 * (size_t) n, nullptr) == SPLAHIP_OK);
        for (int world : {1, 3}) {
            // contiguous row blocks balanced by nnz; rank p's slice of every vector under the same cuts, padded to max_rows
            std::vector<uint32_t> cut(world + 1, n);
            cut[0] = 0;
            for (int p = 1; p < world; ++p) cut[p] = (uint32_t) (std::lower_bound(Ap.begin(), Ap.end(), (uint32_t) ((uint64_t) nnz * p / world)) - Ap.begin());
            uint32_t max_rows = 0;
            for (int p = 0; p < world; ++p) max_rows = std::max(max_rows, cut[p + 1] - cut[p]);
            auto padded = [&](uint32_t col) {// global column -> owner * max_rows + offset
                const int p = (int) (std::upper_bound(cut.begin(), cut.end(), col) - cut.begin()) - 1;
                return world > 1 ? (uint32_t) p * max_rows + (col - cut[p]) : col;
            };
            const uint32_t       n_pad = world > 1 ? (uint32_t) world * max_rows : n;
            std::vector<int32_t> full(n_pad, c.init == INF ? INF : 0);
            for (uint32_t j = 0; j < n; ++j) full[padded(j)] = (*c.hvec)[j];
            std::vector<int32_t> got(n, -12345);
            for (int p = 0; p < world; ++p) {
                const uint32_t lo = cut[p], hi = cut[p + 1], nl = hi - lo, e0 = Ap[lo], nz = Ap[hi] - e0;
                std::vector<uint32_t> bAp(nl + 1), bAj(nz);
                std::vector<int32_t>  bAx(Ax.begin() + e0, Ax.begin() + e0 + nz), vloc(max_rows, full.empty() ? 0 : (c.init == INF ? INF : 0));
                for (uint32_t i = 0; i <= nl; ++i) bAp[i] = Ap[lo + i] - e0;
                for (uint32_t k = 0; k < nz; ++k) bAj[k] = padded(Aj[e0 + k]);
                for (uint32_t i = 0; i < nl; ++i) vloc[i] = (*c.hvec)[lo + i];
                uint32_t *    dbAp = upload(bAp), *dbAj = upload(bAj);
                int32_t *     dbAx = upload(bAx), *dvloc = upload(vloc), *dfull = nullptr, *dbr = upload(std::vector<int32_t>(std::max(nl, 1u), -7));
                int32_t*      dbm  = upload(std::vector<int32_t>(mask.begin() + lo, mask.begin() + hi));
                splahip_shard sh   = nullptr;
                CHECK(dbAp && dbAj && dbAx && dvloc && dbr && dbm);
                CHECK(splahip_shard_create(&sh, world, p, nl, max_rows, nz, dbAp, dbAj, dbAx, nullptr) == SPLAHIP_OK);
                // the exchange: dense (RCCL communicator at world 1; the gathered vector copied in for the simulated ranks), and — for the frontier-like
                // vector — the sparse exchange's own kernels (pack on every rank, messages concatenated as the all-gather would, scatter)
                if (c.vec == dvsp) {
                    std::vector<int32_t> zeros(n_pad, 0);
                    dfull = upload(zeros);
                    std::vector<splahip_sx> sx(world);
                    std::vector<void*>      send(world), recv(world);
                    size_t                  words = 0;
                    for (int q = 0; q < world; ++q) {
                        CHECK(splahip_dist_sx_create(&sx[q], world, q, max_rows, max_rows) == SPLAHIP_OK);
                        std::vector<int32_t> vq(max_rows, 0);
                        for (uint32_t i = 0; i < cut[q + 1] - cut[q]; ++i) vq[i] = vsp[cut[q] + i];
                        int32_t* dvq = upload(vq);
                        CHECK(splahip_dist_sx_pack(sx[q], dvq, cut[q + 1] - cut[q], 0u, 0, 0, nullptr) == SPLAHIP_OK);
                        CHECK(splahip_dist_sx_buffers(sx[q], 0, 0, &send[q], &recv[q], &words) == SPLAHIP_OK);
                        CHECK(splahip_stream_sync(nullptr) == SPLAHIP_OK);
                        splahip_free(dvq);
                    }
                    for (int q = 0; q < world; ++q) CHECK(splahip_memcpy_d2d(static_cast<uint32_t*>(recv[p]) + words * q, send[q], 4 * words, nullptr) == SPLAHIP_OK);
                    CHECK(splahip_dist_sx_scatter(sx[p], dfull, 0u, 0, 0, 1u, nullptr) == SPLAHIP_OK);
                    int over = 1;
                    CHECK(splahip_dist_sx_overflowed(sx[p], &over) == SPLAHIP_OK && over == 0);
                    for (int q = 0; q < world; ++q) splahip_dist_sx_destroy(sx[q]);
                } else
                    dfull = upload(full);
                CHECK(dfull);
                if (world == 1 && splahip_dist_available()) {// the whole product through the communicator (all-gather on its own stream || local part)
                    unsigned char id[128];
                    splahip_comm  comm = nullptr;
                    int           ov   = -1;
                    CHECK(splahip_dist_get_unique_id(id) == SPLAHIP_OK && splahip_dist_comm_create(&comm, 1, 0, id) == SPLAHIP_OK);
                    CHECK(splahip_dist_mxv_masked(comm, sh, SPLAHIP_INT, dbm, dvloc, dfull, dbr, c.mul, c.add, c.sel, (uint32_t) c.init, 0, 0, &ov, nullptr) == SPLAHIP_OK);
                    CHECK(splahip_stream_sync(nullptr) == SPLAHIP_OK && ov == 1);
                    splahip_dist_comm_destroy(comm);
                } else {
                    CHECK(splahip_shard_split_serves(sh, SPLAHIP_INT, c.add, 0) == 1);
                    CHECK(splahip_shard_mxv_local(sh, SPLAHIP_INT, dbm, dvloc, dbr, c.mul, c.add, c.sel, (uint32_t) c.init, nullptr) == SPLAHIP_OK);
                    CHECK(splahip_shard_mxv_remote(sh, SPLAHIP_INT, dbm, dfull, dbr, c.mul, c.add, c.sel, nullptr) == SPLAHIP_OK);
                }
                if (nl) CHECK(splahip_memcpy_d2h(got.data() + lo, dbr, 4 * (size_t) nl, nullptr) == SPLAHIP_OK);
                splahip_shard_destroy(sh);
                for (void* q : {(void*) dbAp, (void*) dbAj, (void*) dbAx, (void*) dvloc, (void*) dfull, (void*) dbr, (void*) dbm}) splahip_free(q);
            }
            for (uint32_t i = 0; i < n; ++i)
                if (got[i] != want[i]) {
                    std::fprintf(stderr, "world %d op_add %d row %u: got %d want %d\n", world, c.add, i, got[i], want[i]);
                    return 1;
                }
        }
        splahip_free(dr);
    }
    splahip_csr_plan_destroy(plan);
    std::printf("DIST C-ABI OK\n");
    return 0;
}
