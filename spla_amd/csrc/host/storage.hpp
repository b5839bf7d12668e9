// storage.hpp — typed containers, their storage "decorations" and the lazy format-validity protocol.
//
// Behaviour follows the reference:
//   * an object owns one slot per format plus a valid-bitset (reference src/core/tdecoration.hpp:69-103);
//   * validate_rw(f)  makes f valid, converting from any valid format along the shortest path of registered
//                     converters (breadth-first search, reference src/storage/storage_manager.hpp:127-204);
//     validate_rwd(f) = validate_rw(f) then every other format becomes invalid;
//     validate_wd(f)  = construct + discard f's content, only f stays valid (reference :211-222);
//   * converter set and their semantics: reference src/storage/storage_manager_{vector,matrix}.hpp and
//     src/cpu/cpu_format_*.hpp (Dok->Coo sorts by index, Dense->Dok drops entries equal to the fill value,
//     Coo->Csr / Coo->Lil trust the caller's row order, `values` counts stored entries);
//   * the accelerator twins AccDense / AccCoo / AccCsr (+ AccCsc: CSR of the transpose, used by vxm's pull form) hold
//     HIP device buffers and are registered only when the HIP accelerator is active.
// The code is this repo's own; only the observable behaviour is shared with the reference.
#pragma once

#include "internal.hpp"

#include <type_traits>
#include <spla_hip.h>

#include <algorithm>
#include <array>
#include <bitset>
#include <cassert>
#include <cstring>
#include <numeric>
#include <queue>
#include <random>
#include <unordered_map>

namespace spla {

    // ---- device buffer (RAII over the kernel-level C ABI allocator) ------------------------------------------------------
    class DeviceBuffer {
    public:
        DeviceBuffer() = default;
        ~DeviceBuffer() { release(); }
        DeviceBuffer(const DeviceBuffer&)            = delete;
        DeviceBuffer& operator=(const DeviceBuffer&) = delete;

        void reserve(std::size_t words) {// contents are NOT preserved
            if (words <= m_words) return;
            release();
            void* p = nullptr;
            if (splahip_malloc(&p, words * 4) != SPLAHIP_OK) throw std::runtime_error(std::string("device allocation failed: ") + splahip_last_error());
            m_ptr   = p;
            m_words = words;
        }
        void release() {
            if (m_ptr) splahip_free(m_ptr);
            m_ptr   = nullptr;
            m_words = 0;
        }
        void upload(const void* host, std::size_t words) {
            reserve(words);
            if (splahip_memcpy_h2d(m_ptr, host, words * 4, nullptr) != SPLAHIP_OK) throw std::runtime_error(std::string("h2d copy failed: ") + splahip_last_error());
        }
        void download(void* host, std::size_t words) const {
            if (splahip_memcpy_d2h(host, m_ptr, words * 4, nullptr) != SPLAHIP_OK) throw std::runtime_error(std::string("d2h copy failed: ") + splahip_last_error());
        }
        void adopt(void* device_ptr, std::size_t words) {// take ownership of a buffer the kernel library allocated
            release();
            m_ptr   = device_ptr;
            m_words = words;
        }
        void*       ptr() { return m_ptr; }
        const void* ptr() const { return m_ptr; }
        template<typename U>
        U* as() { return static_cast<U*>(m_ptr); }
        std::size_t capacity() const { return m_words; }

    private:
        void*       m_ptr   = nullptr;
        std::size_t m_words = 0;
    };

    inline void hip_check(int status, const char* what) {
        if (status != SPLAHIP_OK) throw std::runtime_error(std::string(what) + ": " + splahip_last_error());
    }

    // ---- decorations -------------------------------------------------------------------------------------------------------
    template<typename T>
    struct Decoration : RefCnt {
        uint values = 0;// stored entries
    };

    template<typename T>
    using Reduce = std::function<T(T accum, T added)>;
    template<typename T>
    inline Reduce<T> last_wins() {
        return [](T, T added) { return added; };
    }

    template<typename T>
    struct CpuDokVec final : Decoration<T> {
        std::unordered_map<uint, T> Ax;
        Reduce<T>                   reduce = last_wins<T>();
    };
    template<typename T>
    struct CpuDenseVec final : Decoration<T> {
        std::vector<T> Ax;
    };
    template<typename T>
    struct CpuCooVec final : Decoration<T> {
        std::vector<uint> Ai;
        std::vector<T>    Ax;
    };
    template<typename T>
    struct HipDenseVec final : Decoration<T> {
        DeviceBuffer Ax;// n_rows words
    };
    template<typename T>
    struct HipCooVec final : Decoration<T> {
        DeviceBuffer Ai, Ax;// `values` entries, index-sorted
    };

    template<typename T>
    struct CpuLil final : Decoration<T> {
        using Entry = std::pair<uint, T>;
        std::vector<std::vector<Entry>> Ar;
        Reduce<T>                       reduce = last_wins<T>();
    };
    template<typename T>
    struct CpuDok final : Decoration<T> {
        std::unordered_map<std::uint64_t, T> Ax;// key = row << 32 | col
        Reduce<T>                            reduce = last_wins<T>();
        static std::uint64_t                 key(uint i, uint j) { return (std::uint64_t(i) << 32) | j; }
    };
    template<typename T>
    struct CpuCoo final : Decoration<T> {
        std::vector<uint> Ai, Aj;
        std::vector<T>    Ax;
    };
    template<typename T>
    struct CpuCsr final : Decoration<T> {
        std::vector<uint> Ap, Aj;
        std::vector<T>    Ax;
    };
    // row_bits of the column-ordered-blocks copy (spla_hip.h: splahip_csr_plan_attach_cob): 0 = the library's own range size for every type
    // (20224 slots per range since round 3: a double slice of that many rows fills a CU's LDS, two 4-byte slices share it)
    template<typename T>
    inline uint32_t cob_row_bits(uint64_t) {
        return 0u;
    }

    // device CSR + its work decomposition (splahip_plan); used for AccCsr (the matrix) and AccCsc (its transpose)
    template<typename T>
    struct HipCsr final : Decoration<T> {
        DeviceBuffer Ap, Aj, Ax;
        splahip_plan plan      = nullptr;
        unsigned     mxv_calls = 0;    // mxv / pull-vxm calls without early exit served by this plan (algos_hip.cpp: when the copies of the matrix get built)
        bool         cob_tried = false;// the column-ordered-blocks copy has been asked for (it may have been declined: hypersparse matrix)
        bool         sell_tried = false;
        unsigned     sell_calls = 0;// products that would run on the sliced-ELL copy (splahip_mxv_prefers_sell)
        bool         prepared  = false;// Matrix::set_format(AccCsr / AccCsc): build every copy at the next chance instead of renting first
        ~HipCsr() override {
            if (plan) splahip_csr_plan_destroy(plan);
        }
        void drop_plan() {
            if (plan) splahip_csr_plan_destroy(plan);
            plan      = nullptr;
            mxv_calls = 0, sell_calls = 0;
            cob_tried = sell_tried = false;
        }
    };

    // ---- storage + format graph ------------------------------------------------------------------------------------------------
    template<typename T, int N>
    struct Storage {
        std::array<ref_ptr<Decoration<T>>, N> slots;
        std::bitset<N>                        valid;
        uint                                  n_rows = 0, n_cols = 0;
        T                                     fill   = T();

        template<class D>
        D* get(int slot) { return static_cast<D*>(slots[slot].get()); }
        void invalidate() { valid.reset(); }
    };

    template<typename T, int N>
    class FormatGraph {
    public:
        using S  = Storage<T, N>;
        using Fn = std::function<void(S&)>;

        void set_constructor(int f, Fn fn) { m_ctor[f] = std::move(fn); }
        void set_validator(int f, Fn fn) { m_validator[f] = std::move(fn); }// prepares an object that holds nothing
        void set_discard(int f, Fn fn) { m_discard[f] = std::move(fn); }    // empties f before a write-discard
        void add_converter(int from, int to, Fn fn) { m_edges[from].push_back({to, std::move(fn)}); }
        bool has_format(int f) const { return static_cast<bool>(m_ctor[f]); }

        void ctor(int f, S& s) const {
            if (!s.slots[f]) {
                if (!m_ctor[f]) throw std::runtime_error("format " + std::to_string(f) + " is not available in this build / without the accelerator");
                m_ctor[f](s);
            }
        }

        void rw(int f, S& s) const {
            if (s.valid.test(f)) return;
            if (s.valid.none()) {
                ctor(f, s);
                if (m_validator[f]) m_validator[f](s);
                s.valid.set(f);
                return;
            }
            // breadth-first search from every valid format to f
            std::array<int, N> prev;
            std::array<int, N> via;
            prev.fill(-2);
            std::queue<int> q;
            for (int i = 0; i < N; ++i)
                if (s.valid.test(i)) {
                    prev[i] = -1;
                    q.push(i);
                }
            while (prev[f] == -2) {
                if (q.empty()) throw std::runtime_error("no conversion path to format " + std::to_string(f));
                const int u = q.front();
                q.pop();
                for (std::size_t e = 0; e < m_edges[u].size(); ++e) {
                    const int to = m_edges[u][e].first;
                    if (prev[to] == -2 && m_ctor[to]) {
                        prev[to] = u;
                        via[to]  = static_cast<int>(e);
                        q.push(to);
                    }
                }
            }
            std::vector<int> path;
            for (int cur = f; prev[cur] != -1; cur = prev[cur]) path.push_back(cur);
            for (auto it = path.rbegin(); it != path.rend(); ++it) {
                const int to = *it, from = prev[to];
                ctor(to, s);
                m_edges[from][via[to]].second(s);
                s.valid.set(to);
            }
        }

        void rwd(int f, S& s) const {
            rw(f, s);
            s.valid.reset();
            s.valid.set(f);
        }

        void wd(int f, S& s) const {
            ctor(f, s);
            if (m_discard[f]) m_discard[f](s);
            s.valid.reset();
            s.valid.set(f);
        }

    private:
        std::array<Fn, N>                               m_ctor, m_validator, m_discard;
        std::array<std::vector<std::pair<int, Fn>>, N> m_edges;
    };

    constexpr int kVecFormats = static_cast<int>(FormatVector::Count);
    constexpr int kMatFormats = static_cast<int>(FormatMatrix::Count);

    template<typename T>
    const FormatGraph<T, kVecFormats>& vector_formats();// host/objects.cpp
    template<typename T>
    const FormatGraph<T, kMatFormats>& matrix_formats();

    // ---- typed vector -----------------------------------------------------------------------------------------------------------
    template<typename T>
    class TVector final : public Vector {
    public:
        using S = Storage<T, kVecFormats>;
        explicit TVector(uint n_rows) { m_s.n_rows = n_rows, m_s.n_cols = 1; }

        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        uint               get_n_rows() override { return m_s.n_rows; }
        ref_ptr<Type>      get_type() override { return TypeTraits<T>::type(); }

        Status set_format(FormatVector format) override { return guarded([&] { validate_rw(format); }); }
        Status set_fill_value(const ref_ptr<Scalar>& value) override {
            if (!value) return Status::InvalidArgument;
            m_s.invalidate();// reference src/core/tvector.hpp:131-140
            m_s.fill = scalar_as(value);
            return Status::Ok;
        }
        Status set_reduce(ref_ptr<OpBinary> resolve_duplicates) override {
            auto op = resolve_duplicates.template cast<TOpBinary<T>>();
            if (!op) return Status::InvalidArgument;
            validate_ctor(FormatVector::CpuDok);
            dok()->reduce = op->function;
            return Status::Ok;
        }
        Status set_int(uint i, T_INT v) override { return set(i, static_cast<T>(v)); }
        Status set_uint(uint i, T_UINT v) override { return set(i, static_cast<T>(v)); }
        Status set_float(uint i, T_FLOAT v) override { return set(i, static_cast<T>(v)); }
        Status get_int(uint i, T_INT& v) override { return get(i, v); }
        Status get_uint(uint i, T_UINT& v) override { return get(i, v); }
        Status get_float(uint i, float& v) override { return get(i, v); }

        Status fill_noize(uint seed) override {
            return guarded([&] {
                validate_wd(FormatVector::CpuDense);
                auto&                      Ax = dense()->Ax;
                std::default_random_engine engine(seed);
                if constexpr (std::is_integral<T>::value) {
                    std::uniform_int_distribution<T> dist;
                    for (auto& x : Ax) x = dist(engine);
                } else {
                    std::uniform_real_distribution<T> dist;
                    for (auto& x : Ax) x = dist(engine);
                }
            });
        }
        Status fill_with(const ref_ptr<Scalar>& value) override {
            if (!value) return Status::InvalidArgument;
            return guarded([&] {
                validate_wd(FormatVector::CpuDense);
                std::fill(dense()->Ax.begin(), dense()->Ax.end(), scalar_as(value));
            });
        }
        Status build(const ref_ptr<MemView>& keys, const ref_ptr<MemView>& values) override {
            if (!keys || !values) return Status::InvalidArgument;
            const std::size_t count = keys->get_size() / sizeof(uint);
            if (count * sizeof(uint) != keys->get_size() || count * sizeof(T) != values->get_size()) return Status::InvalidArgument;
            return guarded([&] {
                validate_rwd(FormatVector::CpuCoo);// caller order is kept (reference src/core/tvector.hpp:288-296)
                auto* c = coo();
                c->Ai.resize(count);
                c->Ax.resize(count);
                c->values = static_cast<uint>(count);
                keys->read(0, count * sizeof(uint), c->Ai.data());
                values->read(0, count * sizeof(T), c->Ax.data());
            });
        }
        Status read(ref_ptr<MemView>& keys, ref_ptr<MemView>& values) override {
            return guarded([&] {
                validate_rw(FormatVector::CpuCoo);
                auto* c = coo();
                keys    = MemView::make(c->Ai.data(), c->Ai.size() * sizeof(uint), false);// non-owning views into the object
                values  = MemView::make(c->Ax.data(), c->Ax.size() * sizeof(T), false);
            });
        }
        Status clear() override {
            m_s.invalidate();
            return Status::Ok;
        }

        // format protocol
        void validate_rw(FormatVector f) { vector_formats<T>().rw(static_cast<int>(f), m_s); }
        void validate_rwd(FormatVector f) { vector_formats<T>().rwd(static_cast<int>(f), m_s); }
        void validate_wd(FormatVector f) { vector_formats<T>().wd(static_cast<int>(f), m_s); }
        void validate_ctor(FormatVector f) { vector_formats<T>().ctor(static_cast<int>(f), m_s); }
        bool is_valid(FormatVector f) const { return m_s.valid.test(static_cast<int>(f)); }
        T    get_fill_value() const { return m_s.fill; }
        S&   storage() { return m_s; }

        CpuDokVec<T>*   dok() { return m_s.template get<CpuDokVec<T>>(static_cast<int>(FormatVector::CpuDok)); }
        CpuDenseVec<T>* dense() { return m_s.template get<CpuDenseVec<T>>(static_cast<int>(FormatVector::CpuDense)); }
        CpuCooVec<T>*   coo() { return m_s.template get<CpuCooVec<T>>(static_cast<int>(FormatVector::CpuCoo)); }
        HipDenseVec<T>* acc_dense() { return m_s.template get<HipDenseVec<T>>(static_cast<int>(FormatVector::AccDense)); }
        HipCooVec<T>*   acc_coo() { return m_s.template get<HipCooVec<T>>(static_cast<int>(FormatVector::AccCoo)); }

    private:
        static T scalar_as(const ref_ptr<Scalar>& s) {
            if constexpr (std::is_same<T, T_INT>::value) return s->as_int();
            else if constexpr (std::is_same<T, T_UINT>::value)
                return s->as_uint();
            else
                return s->as_float();
        }
        template<class F>
        Status guarded(F&& f) {
            try {
                f();
                return Status::Ok;
            } catch (const std::exception& e) {
                SPLA_LOG(Status::Error, e.what());
                return Status::Error;
            }
        }
        // element writes go straight to the dense array while it is the valid format, else to the DOK map with the
        // object's duplicate-resolution op (reference src/core/tvector.hpp:158-193)
        Status set(uint i, T v) {
            if (i >= m_s.n_rows) return Status::InvalidArgument;
            return guarded([&] {
                if (is_valid(FormatVector::CpuDense)) {
                    validate_rwd(FormatVector::CpuDense);
                    dense()->Ax[i] = v;
                    return;
                }
                validate_rwd(FormatVector::CpuDok);
                auto* d  = dok();
                auto  it = d->Ax.find(i);
                if (it != d->Ax.end()) it->second = d->reduce(it->second, v);
                else {
                    d->Ax.emplace(i, v);
                    d->values += 1;
                }
            });
        }
        template<typename U>
        Status get(uint i, U& out) {
            return guarded([&] {
                validate_rw(FormatVector::CpuDok);
                out     = static_cast<U>(m_s.fill);
                auto it = dok()->Ax.find(i);
                if (it != dok()->Ax.end()) out = static_cast<U>(it->second);
            });
        }

        S           m_s;
        std::string m_label;
    };

    // ---- typed matrix -----------------------------------------------------------------------------------------------------------
    template<typename T>
    class TMatrix final : public Matrix {
    public:
        using S = Storage<T, kMatFormats>;
        TMatrix(uint n_rows, uint n_cols) { m_s.n_rows = n_rows, m_s.n_cols = n_cols; }

        void               set_label(std::string label) override { m_label = std::move(label); }
        const std::string& get_label() const override { return m_label; }
        uint               get_n_rows() override { return m_s.n_rows; }
        uint               get_n_cols() override { return m_s.n_cols; }
        ref_ptr<Type>      get_type() override { return TypeTraits<T>::type(); }

        // reference include/spla/matrix.hpp:51-65: "set_format ... instructs the matrix to keep this format valid".  For the accelerator formats it is
        // also the caller's way to say "prepare now": the plan and the column-ordered-blocks copy of the device CSR are built here, before the first product,
        // instead of after the matrix has shown that it keeps being multiplied (algos_hip.cpp: rent-or-buy).  AccCsc: flagged, built with the transpose.
        Status set_format(FormatMatrix format) override {
            return guarded([&] {
                validate_rw(format);
                if (format == FormatMatrix::AccCsr) {
                    auto* a     = acc_csr();
                    a->prepared = true;
                    if (a->values > 0) {
                        if (!a->plan && splahip_csr_plan_create(&a->plan, m_s.n_rows, a->values, a->Ap.template as<uint32_t>(), nullptr) != SPLAHIP_OK)
                            throw std::runtime_error(std::string("csr plan: ") + splahip_last_error());
                        if (!a->cob_tried) {
                            a->cob_tried = true;
                            (void) splahip_csr_plan_attach_cob(a->plan, m_s.n_cols, a->Ap.template as<uint32_t>(), a->Aj.template as<uint32_t>(), a->Ax.ptr(), cob_row_bits<T>(m_s.n_rows), 0, nullptr);
                        }
                        // (the sliced-ELL copy, +12 B per entry, follows at the first product that would use it: algos_hip.cpp, maybe_build_copies)
                    }
                }
            });
        }
        Status set_fill_value(const ref_ptr<Scalar>& value) override {
            if (!value) return Status::InvalidArgument;
            m_s.invalidate();
            if constexpr (std::is_same<T, T_INT>::value) m_s.fill = value->as_int();
            else if constexpr (std::is_same<T, T_UINT>::value)
                m_s.fill = value->as_uint();
            else
                m_s.fill = value->as_float();
            return Status::Ok;
        }
        Status set_reduce(ref_ptr<OpBinary> resolve_duplicates) override {
            auto op = resolve_duplicates.template cast<TOpBinary<T>>();
            if (!op) return Status::InvalidArgument;
            validate_ctor(FormatMatrix::CpuLil);
            lil()->reduce = op->function;
            validate_ctor(FormatMatrix::CpuDok);
            dok()->reduce = op->function;
            return Status::Ok;// (the reference returns InvalidArgument here even on success, src/core/tmatrix.hpp:143-155)
        }
        Status set_int(uint i, uint j, std::int32_t v) override { return set(i, j, static_cast<T>(v)); }
        Status set_uint(uint i, uint j, std::uint32_t v) override { return set(i, j, static_cast<T>(v)); }
        Status set_float(uint i, uint j, float v) override { return set(i, j, static_cast<T>(v)); }
        Status get_int(uint i, uint j, std::int32_t& v) override { return get(i, j, v); }
        Status get_uint(uint i, uint j, std::uint32_t& v) override { return get(i, j, v); }
        Status get_float(uint i, uint j, float& v) override { return get(i, j, v); }

        Status build(const ref_ptr<MemView>& keys1, const ref_ptr<MemView>& keys2, const ref_ptr<MemView>& values) override {
            if (!keys1 || !keys2 || !values) return Status::InvalidArgument;
            const std::size_t count = keys1->get_size() / sizeof(uint);
            if (count * sizeof(uint) != keys1->get_size() || count * sizeof(uint) != keys2->get_size() || count * sizeof(T) != values->get_size())
                return Status::InvalidArgument;
            return guarded([&] {
                validate_rwd(FormatMatrix::CpuCoo);// caller-ordered COO; converters trust it to be row-sorted
                auto* c = coo();
                c->Ai.resize(count);
                c->Aj.resize(count);
                c->Ax.resize(count);
                c->values = static_cast<uint>(count);
                keys1->read(0, count * sizeof(uint), c->Ai.data());
                keys2->read(0, count * sizeof(uint), c->Aj.data());
                values->read(0, count * sizeof(T), c->Ax.data());
            });
        }
        Status read(ref_ptr<MemView>& keys1, ref_ptr<MemView>& keys2, ref_ptr<MemView>& values) override {
            return guarded([&] {
                validate_rw(FormatMatrix::CpuCoo);
                auto* c = coo();
                keys1   = MemView::make(c->Ai.data(), c->Ai.size() * sizeof(uint), false);
                keys2   = MemView::make(c->Aj.data(), c->Aj.size() * sizeof(uint), false);
                values  = MemView::make(c->Ax.data(), c->Ax.size() * sizeof(T), false);
            });
        }
        Status clear() override {
            m_s.invalidate();
            return Status::Ok;
        }

        void validate_rw(FormatMatrix f) { matrix_formats<T>().rw(static_cast<int>(f), m_s); }
        void validate_rwd(FormatMatrix f) { matrix_formats<T>().rwd(static_cast<int>(f), m_s); }
        void validate_wd(FormatMatrix f) { matrix_formats<T>().wd(static_cast<int>(f), m_s); }
        void validate_ctor(FormatMatrix f) { matrix_formats<T>().ctor(static_cast<int>(f), m_s); }
        bool is_valid(FormatMatrix f) const { return m_s.valid.test(static_cast<int>(f)); }
        T    get_fill_value() const { return m_s.fill; }
        S&   storage() { return m_s; }

        CpuLil<T>* lil() { return m_s.template get<CpuLil<T>>(static_cast<int>(FormatMatrix::CpuLil)); }
        CpuDok<T>* dok() { return m_s.template get<CpuDok<T>>(static_cast<int>(FormatMatrix::CpuDok)); }
        CpuCoo<T>* coo() { return m_s.template get<CpuCoo<T>>(static_cast<int>(FormatMatrix::CpuCoo)); }
        CpuCsr<T>* csr() { return m_s.template get<CpuCsr<T>>(static_cast<int>(FormatMatrix::CpuCsr)); }
        HipCsr<T>* acc_csr() { return m_s.template get<HipCsr<T>>(static_cast<int>(FormatMatrix::AccCsr)); }
        HipCsr<T>* acc_csc() { return m_s.template get<HipCsr<T>>(static_cast<int>(FormatMatrix::AccCsc)); }

    private:
        template<class F>
        Status guarded(F&& f) {
            try {
                f();
                return Status::Ok;
            } catch (const std::exception& e) {
                SPLA_LOG(Status::Error, e.what());
                return Status::Error;
            }
        }
        // element writes always go to the LIL rows: column-sorted insertion, duplicates resolved by `reduce`
        // (reference src/core/tmatrix.hpp:158-175, src/cpu/cpu_format_lil.hpp:54-73)
        Status set(uint i, uint j, T v) {
            if (i >= m_s.n_rows || j >= m_s.n_cols) return Status::InvalidArgument;
            return guarded([&] {
                validate_rwd(FormatMatrix::CpuLil);
                auto* l   = lil();
                auto& row = l->Ar[i];
                auto  pos = std::lower_bound(row.begin(), row.end(), j, [](const typename CpuLil<T>::Entry& e, uint col) { return e.first < col; });
                if (pos != row.end() && pos->first == j) pos->second = l->reduce(pos->second, v);
                else {
                    row.insert(pos, {j, v});
                    l->values += 1;
                }
            });
        }
        template<typename U>
        Status get(uint i, uint j, U& out) {
            return guarded([&] {
                validate_rw(FormatMatrix::CpuDok);
                out     = static_cast<U>(m_s.fill);
                auto it = dok()->Ax.find(CpuDok<T>::key(i, j));
                if (it != dok()->Ax.end()) out = static_cast<U>(it->second);
            });
        }

        S           m_s;
        std::string m_label;
    };

}// namespace spla
