// User-defined ops through the C++ operator API (the four hot-path operators and, since round 4, the vector / matrix operations around them), the way a reference user writes them (reference tests/test_op.cpp:56-59 makes the op,
// tests/test_mxv.cpp:33-89 / test_vxm.cpp / test_mxm.cpp / test_mxmT.cpp are the products): exec_mxv_masked, exec_vxm_masked, exec_mxm and
// exec_mxmT_masked with OpBinary::make_float / OpSelect::make_float ops run ON THE DEVICE — the body text is compiled at run time
// (spla_amd/csrc/kernels/jit.hip) — and must equal the host evaluation of the same std::function in the reference's fold order
// (src/cpu/cpu_mxv.hpp:88-103, cpu_vxm.hpp:92-125, cpu_mxm.hpp:85-114, cpu_mxmT_masked.hpp:87-125).  The ops are NOT associative and their
// products are inexact in fp32 (a * b + 1 would contract to a fused multiply-add if the device compiler were allowed to), so any other order
// or any contraction shows.
// Built by oracle/Makefile (target usertests) against include/spla.hpp + libspla_amd.so; run by tests/test_user_ops_gpu.py.
#include <spla.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main() {
    spla::Library::get()->set_accelerator(spla::AcceleratorType::Hip);

    // ---- fp32: custom_plus as op_add (not associative: the device must fold left to right like the reference), built-in MULT ----
    auto custom_plus = spla::OpBinary::make_float("custom_plus", "(float a, float b) { return 0.25f * a + 0.75f * b; }",
                                                  [](float a, float b) { return 0.25f * a + 0.75f * b; });
    auto custom_mult = spla::OpBinary::make_float("custom_mult", "(float a, float b) { return a * b + 1.0f; }", [](float a, float b) { return a * b + 1.0f; });
    auto custom_sel  = spla::OpSelect::make_float("custom_sel", "(float a) { return a > 0.5f; }", [](float a) { return a > 0.5f; });
    const spla::uint M = 257, N = 300;
    auto             A = spla::Matrix::make(M, N, spla::FLOAT);
    auto             v = spla::Vector::make(N, spla::FLOAT);
    auto             mask = spla::Vector::make(M, spla::FLOAT);
    auto             r    = spla::Vector::make(M, spla::FLOAT);
    std::vector<std::vector<std::pair<spla::uint, float>>> rows(M);
    std::vector<float>                                     hv(N), hm(M);
    unsigned                                               seed = 12345;
    auto                                                   rnd  = [&]() { return (seed = seed * 1664525u + 1013904223u) >> 8; };
    for (spla::uint j = 0; j < N; ++j) {
        hv[j] = (rnd() % 64) / 16.0f;
        v->set_float(j, hv[j]);
    }
    for (spla::uint i = 0; i < M; ++i) {
        hm[i] = (rnd() % 4) / 4.0f;
        mask->set_float(i, hm[i]);
        const unsigned deg = i == 7 ? 250 : rnd() % 9;// one long row
        spla::uint     j   = rnd() % 3;
        for (unsigned d = 0; d < deg && j < N; ++d, j += 1 + rnd() % (i == 7 ? 1 : 30)) {
            const float x = (rnd() % 32) / 8.0f;
            rows[i].push_back({j, x});
            A->set_float(i, j, x);
        }
    }
    for (int early = 0; early < 2; ++early) {
        auto desc = spla::Descriptor::make();
        desc->set_early_exit(early != 0);
        const float init = 0.5f;
        spla::Status st = spla::exec_mxv_masked(r, mask, A, v, custom_mult, custom_plus, custom_sel, spla::Scalar::make_float(init), desc);
        CHECK(st == spla::Status::Ok);
        for (spla::uint i = 0; i < M; ++i) {
            float want = init;
            if (hm[i] > 0.5f)
                for (auto& e : rows[i]) {
                    want = 0.25f * want + 0.75f * (e.second * hv[e.first] + 1.0f);
                    if (early && want != init) break;
                }
            float got = -1.0f;
            CHECK(r->get_float(i, got) == spla::Status::Ok);
            if (got != want) {
                std::fprintf(stderr, "row %u early=%d: got %.9g want %.9g\n", i, early, got, want);
                return 1;
            }
        }
    }

    // ---- the other three operators with the same non-associative, inexact ops (values with long mantissas: 1/3-ish steps) ----
    {
        auto fplus = [](float a, float b) { return 0.25f * a + 0.75f * b; };
        auto fmult = [](float a, float b) { return a * b + 1.0f; };
        auto fsel  = [](float a) { return a > 0.5f; };
        const spla::uint P = 60, Q = 70, K = 50;// A: P x K, B: K x Q (mxm); Bt: Q x K (mxmT); vxm: u (P) x A (P x K)
        std::vector<std::vector<std::pair<spla::uint, float>>> ra(P), rb(K), rbt(Q);
        auto val = [&]() { return 0.1f + (rnd() % 97) / 29.0f; };
        auto A2 = spla::Matrix::make(P, K, spla::FLOAT), B2 = spla::Matrix::make(K, Q, spla::FLOAT), Bt = spla::Matrix::make(Q, K, spla::FLOAT);
        for (spla::uint i = 0; i < P; ++i)
            for (spla::uint k = rnd() % 4; k < K; k += 1 + rnd() % 5) {
                const float x = val();
                ra[i].push_back({k, x});
                A2->set_float(i, k, x);
            }
        for (spla::uint k = 0; k < K; ++k)
            for (spla::uint j = rnd() % 3; j < Q; j += 1 + rnd() % 4) {
                const float x = val();
                rb[k].push_back({j, x});
                B2->set_float(k, j, x);
            }
        for (spla::uint j = 0; j < Q; ++j)
            for (spla::uint k = rnd() % 4; k < K; k += 1 + rnd() % 5) {
                const float x = val();
                rbt[j].push_back({k, x});
                Bt->set_float(j, k, x);
            }
        const float init = 0.375f;
        // mxm: R[i, j] = fold over k in A[i, :] (row order) of plus(., mult(a, b)) from init; kept iff != init
        {
            auto R = spla::Matrix::make(P, Q, spla::FLOAT);
            CHECK(spla::exec_mxm(R, A2, B2, custom_mult, custom_plus, spla::Scalar::make_float(init)) == spla::Status::Ok);
            for (spla::uint i = 0; i < P; ++i) {
                std::vector<float> acc(Q, init);
                for (auto& ea : ra[i])
                    for (auto& eb : rb[ea.first]) acc[eb.first] = fplus(acc[eb.first], fmult(ea.second, eb.second));
                for (spla::uint j = 0; j < Q; ++j) {
                    float got = -7.0f;
                    R->get_float(i, j, got);
                    const float want = acc[j] != init ? acc[j] : 0.0f;// an absent entry reads as the fill value
                    if (got != want) {
                        std::fprintf(stderr, "mxm (%u, %u): got %.9g want %.9g\n", i, j, got, want);
                        return 1;
                    }
                }
            }
        }
        // mxmT_masked: for every stored mask entry (i, j, m) with select(m): merge of the sorted rows A[i, :] and Bt[j, :]
        {
            auto Mk = spla::Matrix::make(P, Q, spla::FLOAT);
            std::vector<std::vector<std::pair<spla::uint, float>>> rm(P);
            for (spla::uint i = 0; i < P; ++i)
                for (spla::uint j = rnd() % 3; j < Q; j += 1 + rnd() % 3) {
                    const float m = (rnd() % 4) / 4.0f + 0.125f;
                    rm[i].push_back({j, m});
                    Mk->set_float(i, j, m);
                }
            auto R = spla::Matrix::make(P, Q, spla::FLOAT);
            CHECK(spla::exec_mxmT_masked(R, Mk, A2, Bt, custom_mult, custom_plus, custom_sel, spla::Scalar::make_float(init)) == spla::Status::Ok);
            for (spla::uint i = 0; i < P; ++i) {
                std::vector<float> want(Q, 0.0f);
                for (auto& em : rm[i]) {
                    float r0 = init;
                    if (fsel(em.second)) {
                        size_t ka = 0, kb = 0;
                        const auto &a = ra[i], &b = rbt[em.first];
                        while (ka < a.size() && kb < b.size()) {
                            if (a[ka].first == b[kb].first) r0 = fplus(r0, fmult(a[ka].second, b[kb].second)), ++ka, ++kb;
                            else if (a[ka].first < b[kb].first) ++ka;
                            else ++kb;
                        }
                    }
                    if (r0 != init) want[em.first] = r0;
                }
                for (spla::uint j = 0; j < Q; ++j) {
                    float got = -7.0f;
                    R->get_float(i, j, got);
                    if (got != want[j]) {
                        std::fprintf(stderr, "mxmT (%u, %u): got %.9g want %.9g\n", i, j, got, want[j]);
                        return 1;
                    }
                }
            }
        }
        // vxm: for every stored entry (i, u_i) of u in stored order, for (k, x) in A[i, :] with select(mask[k]): the first product of column k
        // initialises, later ones fold.  Twice: u set element by element (stored index-sorted), and u built from an UNSORTED list (stored as given)
        for (int unsorted = 0; unsorted < 2; ++unsorted) {
            std::vector<spla::uint> ui;
            std::vector<float>      ux;
            for (spla::uint i = rnd() % 2; i < P; i += 1 + rnd() % 3) ui.push_back(i), ux.push_back(val());
            auto u = spla::Vector::make(P, spla::FLOAT);
            if (unsorted) {
                for (size_t t = ui.size(); t > 1; --t) {
                    const size_t o = rnd() % t;
                    std::swap(ui[t - 1], ui[o]);
                    std::swap(ux[t - 1], ux[o]);
                }
                CHECK(u->build(spla::MemView::make(ui.data(), ui.size() * sizeof(spla::uint)), spla::MemView::make(ux.data(), ux.size() * sizeof(float))) == spla::Status::Ok);
            } else
                for (size_t t = 0; t < ui.size(); ++t) u->set_float(ui[t], ux[t]);
            auto               mk2 = spla::Vector::make(K, spla::FLOAT);
            std::vector<float> hm2(K);
            for (spla::uint k = 0; k < K; ++k) {
                hm2[k] = (rnd() % 4) / 4.0f + 0.125f;
                mk2->set_float(k, hm2[k]);
            }
            auto w = spla::Vector::make(K, spla::FLOAT);
            CHECK(spla::exec_vxm_masked(w, mk2, u, A2, custom_mult, custom_plus, custom_sel, spla::Scalar::make_float(init)) == spla::Status::Ok);
            std::vector<float> want(K, 0.0f);
            std::vector<char>  has(K, 0);
            for (size_t t = 0; t < ui.size(); ++t)
                for (auto& e : ra[ui[t]]) {
                    if (!fsel(hm2[e.first])) continue;
                    const float p = fmult(ux[t], e.second);
                    want[e.first] = has[e.first] ? fplus(want[e.first], p) : p;
                    has[e.first]  = 1;
                }
            for (spla::uint k = 0; k < K; ++k) {
                float got = -7.0f;
                w->get_float(k, got);
                if (got != want[k]) {
                    std::fprintf(stderr, "vxm (unsorted=%d) column %u: got %.9g want %.9g\n", unsorted, k, got, want[k]);
                    return 1;
                }
            }
        }
    }

    // ---- the vector operations around the hot path with user-defined ops: v_assign_masked, v_eadd_fdb, v_eadd, v_reduce, m_reduce, v_map, m_transpose,
    //      each in its sparse form (operands set element by element) and its dense form (operands prepared with set_format(CpuDense)); reference semantics:
    //      src/cpu/cpu_v_assign.hpp:68-132, cpu_v_eadd_fdb.hpp:71-139, cpu_v_eadd.hpp:70-157, cpu_v_reduce.hpp:69-117, cpu_m_reduce.hpp:134-156, cpu_v_map.hpp:57-68 ----
    {
        auto fop  = [](float a, float b) { return 0.25f * a + 0.75f * b; };// not associative, not commutative
        auto fsel = [](float a) { return a > 0.5f; };
        auto fun  = [](float a) { return a * a + 1.0f; };
        auto custom_un = spla::OpUnary::make_float("custom_un", "(float a) { return a * a + 1.0f; }", fun);
        const spla::uint L = 1500;
        for (int dense = 0; dense < 2; ++dense) {
            std::vector<float> hu(L, 0.0f), hw(L, 0.0f), hk(L, 0.0f), hr(L, 0.0f);
            std::vector<char>  su(L, 0), sw(L, 0), sk(L, 0);
            auto u = spla::Vector::make(L, spla::FLOAT), w = spla::Vector::make(L, spla::FLOAT), mk = spla::Vector::make(L, spla::FLOAT), rr = spla::Vector::make(L, spla::FLOAT);
            for (spla::uint i = 0; i < L; ++i) {
                if (rnd() % 3 != 0) hu[i] = 0.125f + (rnd() % 61) / 7.0f, su[i] = 1, u->set_float(i, hu[i]);
                if (rnd() % 2 != 0) hw[i] = 0.125f + (rnd() % 53) / 11.0f, sw[i] = 1, w->set_float(i, hw[i]);
                if (rnd() % 4 != 0) hk[i] = (rnd() % 5) / 4.0f, sk[i] = 1, mk->set_float(i, hk[i]);
                hr[i] = (rnd() % 19) / 3.0f;
                rr->set_float(i, hr[i]);
            }
            if (dense) {
                CHECK(u->set_format(spla::FormatVector::CpuDense) == spla::Status::Ok);
                CHECK(w->set_format(spla::FormatVector::CpuDense) == spla::Status::Ok);
                CHECK(mk->set_format(spla::FormatVector::CpuDense) == spla::Status::Ok);
            }
            CHECK(rr->set_format(spla::FormatVector::CpuDense) == spla::Status::Ok);
            auto same = [&](const spla::ref_ptr<spla::Vector>& x, const std::vector<float>& want, const char* what) {
                for (spla::uint i = 0; i < L; ++i) {
                    float got = -9.0f;
                    x->get_float(i, got);
                    if (got != want[i]) {
                        std::fprintf(stderr, "%s (dense=%d) entry %u: got %.9g want %.9g\n", what, dense, i, got, want[i]);
                        return false;
                    }
                }
                return true;
            };
            // v_assign_masked: r[i] = op(r[i], value) where select(mask[i]) — over the stored mask entries (sparse) / over every index, absent = fill 0 (dense)
            {
                const float value = 3.5f;
                CHECK(spla::exec_v_assign_masked(rr, mk, spla::Scalar::make_float(value), custom_plus, custom_sel) == spla::Status::Ok);
                for (spla::uint i = 0; i < L; ++i)
                    if ((dense || sk[i]) && fsel(hk[i])) hr[i] = fop(hr[i], value);
                CHECK(same(rr, hr, "v_assign_masked"));
            }
            // v_eadd_fdb: r[i] = op(r[i], w[i]) over w's stored entries (sparse) / every index (dense); fdb = the changed entries
            {
                auto fdb = spla::Vector::make(L, spla::FLOAT);
                CHECK(spla::exec_v_eadd_fdb(rr, w, fdb, custom_plus) == spla::Status::Ok);
                std::vector<float> hf(L, 0.0f);
                for (spla::uint i = 0; i < L; ++i)
                    if (dense || sw[i]) {
                        const float prev = hr[i], next = fop(prev, hw[i]);
                        hr[i] = next;
                        if (prev != next) hf[i] = next;
                    }
                CHECK(same(rr, hr, "v_eadd_fdb r"));
                CHECK(same(fdb, hf, "v_eadd_fdb fdb"));
            }
            // v_eadd: op(u[i], w[i]) — sparse: over the union of the stored indices with the other side's fill value (0); dense: every index
            {
                auto out = spla::Vector::make(L, spla::FLOAT);
                CHECK(spla::exec_v_eadd(out, u, w, custom_plus) == spla::Status::Ok);
                std::vector<float> ho(L, 0.0f);
                for (spla::uint i = 0; i < L; ++i)
                    if (dense || su[i] || sw[i]) ho[i] = fop(hu[i], hw[i]);
                CHECK(same(out, ho, "v_eadd"));
            }
            // v_reduce: the strict left fold from s over the stored values (sparse) / every entry (dense)
            {
                auto        res  = spla::Scalar::make_float(0.0f);
                const float s0   = 1.75f;
                float       want = s0;
                for (spla::uint i = 0; i < L; ++i)
                    if (dense || su[i]) want = fop(want, hu[i]);
                CHECK(spla::exec_v_reduce(res, spla::Scalar::make_float(s0), u, custom_plus) == spla::Status::Ok);
                if (res->as_float() != want) {
                    std::fprintf(stderr, "v_reduce (dense=%d): got %.9g want %.9g\n", dense, res->as_float(), want);
                    return 1;
                }
            }
            // v_map: op(u[i]) over the stored entries (sparse) / every index (dense)
            {
                auto out = spla::Vector::make(L, spla::FLOAT);
                CHECK(spla::exec_v_map(out, u, custom_un) == spla::Status::Ok);
                std::vector<float> ho(L, 0.0f);
                for (spla::uint i = 0; i < L; ++i)
                    if (dense || su[i]) ho[i] = fun(hu[i]);
                CHECK(same(out, ho, "v_map"));
            }
        }
        // m_reduce (row-major fold of the stored values) and m_transpose with a user-defined unary op, on the matrix of the first test
        {
            auto        res  = spla::Scalar::make_float(0.0f);
            const float s0   = 0.625f;
            float       want = s0;
            for (spla::uint i = 0; i < M; ++i)
                for (auto& e : rows[i]) want = fop(want, e.second);
            CHECK(spla::exec_m_reduce(res, spla::Scalar::make_float(s0), A, custom_plus) == spla::Status::Ok);
            if (res->as_float() != want) {
                std::fprintf(stderr, "m_reduce: got %.9g want %.9g\n", res->as_float(), want);
                return 1;
            }
            auto At = spla::Matrix::make(N, M, spla::FLOAT);
            CHECK(spla::exec_m_transpose(At, A, custom_un) == spla::Status::Ok);
            for (spla::uint i = 0; i < M; ++i)
                for (auto& e : rows[i]) {
                    float got = -9.0f;
                    At->get_float(e.first, i, got);
                    if (got != fun(e.second)) {
                        std::fprintf(stderr, "m_transpose (%u, %u): got %.9g want %.9g\n", e.first, i, got, fun(e.second));
                        return 1;
                    }
                }
        }
        // a body that does not compile: CompilationError from a vector operation too
        auto broken_f = spla::OpBinary::make_float("broken_f", "(float a, float b) { return a +* b; }", [](float a, float b) { return a + b; });
        auto x1 = spla::Vector::make(8, spla::FLOAT), x2 = spla::Vector::make(8, spla::FLOAT), x3 = spla::Vector::make(8, spla::FLOAT);
        x1->set_float(1, 1.0f), x2->set_float(2, 2.0f);
        CHECK(spla::exec_v_eadd(x3, x1, x2, broken_f) == spla::Status::CompilationError);
    }

    // ---- int32: a user-defined op_add beside the built-in PLUS / EQZERO (saturating add: min(a + b, 1000)) ----
    auto sat_add = spla::OpBinary::make_int("sat_add", "(int a, int b) { return a + b > 1000 ? 1000 : a + b; }", [](int a, int b) { return a + b > 1000 ? 1000 : a + b; });
    auto Ai      = spla::Matrix::make(4, 5, spla::INT);
    auto vi      = spla::Vector::make(5, spla::INT);
    auto mi      = spla::Vector::make(4, spla::INT);
    auto ri      = spla::Vector::make(4, spla::INT);
    const int vals[5] = {300, 0, 300, 0, 700};
    for (int j = 0; j < 5; ++j) vi->set_int(j, vals[j]);
    const int mk[4] = {1, 0, 1, 0};
    for (int i = 0; i < 4; ++i) mi->set_int(i, mk[i]);
    Ai->set_int(1, 0, 2), Ai->set_int(1, 2, 1), Ai->set_int(1, 4, 1), Ai->set_int(3, 4, 1), Ai->set_int(0, 1, 5);
    CHECK(spla::exec_mxv_masked(ri, mi, Ai, vi, spla::MULT_INT, sat_add, spla::EQZERO_INT, spla::Scalar::make_int(0)) == spla::Status::Ok);
    int got[4];
    for (int i = 0; i < 4; ++i) CHECK(ri->get_int(i, got[i]) == spla::Status::Ok);
    CHECK(got[0] == 0 && got[1] == 1000 && got[2] == 0 && got[3] == 700);

    // ---- a body that does not compile: the reference's Status::CompilationError, not a crash ----
    auto broken = spla::OpBinary::make_int("broken", "(int a, int b) { return a +* b; }", [](int a, int b) { return a + b; });
    CHECK(spla::exec_mxv_masked(ri, mi, Ai, vi, spla::MULT_INT, broken, spla::EQZERO_INT, spla::Scalar::make_int(0)) == spla::Status::CompilationError);
    std::printf("USER-OPS OK\n");
    return 0;
}
