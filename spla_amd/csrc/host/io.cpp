// io.cpp — MatrixMarket coordinate loader + degree statistics.  Host-side input plumbing with the
// interface of the reference's MtxLoader (include/spla/io.hpp:47-85; behaviour of src/io.cpp:50-233): '%' comment lines, a
// "rows cols entries" line, then one "i j [value]" line per entry of which only the two indices are used; options drop self
// loops, shift the 1-based indices to 0-based and add the reverse of every edge; the result is sorted by (row, column) with
// duplicates removed.  Written from scratch: the file is read in one piece and walked with a pointer.
#include "../../../include/spla.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

namespace spla {

    MtxLoader::MtxLoader(std::string name) : m_name(std::move(name)) {}

    bool MtxLoader::load(std::filesystem::path file_path, bool offset_indices, bool make_undirected, bool remove_loops) {
        m_file_path    = std::move(file_path);
        m_base_is_zero = offset_indices;
        std::ifstream in(m_file_path, std::ios::binary);
        if (!in) {
            std::cerr << "MtxLoader: cannot open " << m_file_path << std::endl;
            return false;
        }
        std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const char* p   = text.c_str();
        const char* end = p + text.size();
        auto        skip_line = [&]() {
            while (p < end && *p != '\n') ++p;
            if (p < end) ++p;
        };
        while (p < end && *p == '%') skip_line();
        char*              q       = nullptr;
        const std::uint64_t rows   = std::strtoull(p, &q, 10);
        const std::uint64_t cols   = std::strtoull(q, &q, 10);
        const std::uint64_t listed = std::strtoull(q, &q, 10);
        p = q;
        skip_line();
        m_n_rows = static_cast<uint>(rows);
        m_n_cols = static_cast<uint>(cols);

        std::vector<std::uint64_t> edges;// (row << 32) | column
        edges.reserve(static_cast<std::size_t>(listed) * (make_undirected ? 2 : 1));
        for (std::uint64_t k = 0; k < listed && p < end; ++k) {
            std::uint64_t i = std::strtoull(p, &q, 10);
            if (q == p) break;// malformed / truncated file: keep what was read
            std::uint64_t j = std::strtoull(q, &q, 10);
            p               = q;
            skip_line();// a value column, if any, is ignored
            if (remove_loops && i == j) continue;
            if (offset_indices) {
                i -= 1;
                j -= 1;
            }
            edges.push_back((i << 32) | j);
            if (make_undirected) edges.push_back((j << 32) | i);
        }
        std::sort(edges.begin(), edges.end());
        edges.erase(std::unique(edges.begin(), edges.end()), edges.end());

        m_Ai.resize(edges.size());
        m_Aj.resize(edges.size());
        for (std::size_t k = 0; k < edges.size(); ++k) {
            m_Ai[k] = static_cast<uint>(edges[k] >> 32);
            m_Aj[k] = static_cast<uint>(edges[k] & 0xffffffffull);
        }
        m_n_values = edges.size();
        std::cout << "MtxLoader: " << m_file_path << ": " << m_n_rows << " x " << m_n_cols << ", " << listed << " listed entries -> " << m_n_values
                  << " edges" << (remove_loops ? ", self loops removed" : "") << (make_undirected ? ", reverse edges added" : "") << std::endl;
        calc_stats();
        output_stats();
        return true;
    }

    bool MtxLoader::save(const std::filesystem::path& file_path, bool stats_only) {
        std::ofstream out(file_path);
        if (!out) return false;
        out << "%%MatrixMarket matrix coordinate pattern general\n";
        out << "% name: " << m_name << "\n% source-file: " << m_file_path << "\n";
        out << "% deg-avg: " << m_deg_avg << "\n% deg-sd: " << m_deg_sd << "\n% deg-min: " << m_deg_min << "\n% deg-max: " << m_deg_max << "\n";
        for (std::size_t g = 0; g < m_deg_distribution.size(); ++g)
            out << "% deg-share " << m_deg_ranges[g] << " .. " << m_deg_ranges[g + 1] << ": " << m_deg_distribution[g] << "\n";
        out << m_n_rows << " " << m_n_cols << " " << m_n_values << "\n";
        if (!stats_only) {
            const uint shift = m_base_is_zero ? 1 : 0;
            for (std::size_t k = 0; k < m_n_values; ++k) out << m_Ai[k] + shift << " " << m_Aj[k] + shift << "\n";
        }
        return static_cast<bool>(out);
    }

    // out-degree statistics: min / max / mean / sample standard deviation (as the reference prints them, src/io.cpp:273-295) and
    // the share of vertices per degree range (ranges = equal-population cuts of the sorted degree sequence)
    void MtxLoader::calc_stats() {
        std::vector<uint> deg(m_n_rows, 0);
        for (uint i : m_Ai) {
            const uint row = m_base_is_zero ? i : i - 1;
            if (row < m_n_rows) deg[row] += 1;
        }
        if (deg.empty()) {
            m_deg_min = m_deg_max = m_deg_avg = m_deg_sd = 0.0;
            m_deg_distribution.clear();
            m_deg_ranges.clear();
            return;
        }
        double sum = 0.0, sum2 = 0.0;
        uint   lo = deg[0], hi = deg[0];
        for (uint d : deg) {
            sum += d;
            sum2 += double(d) * double(d);
            lo = std::min(lo, d);
            hi = std::max(hi, d);
        }
        const double n = static_cast<double>(m_n_rows);
        m_deg_min      = lo;
        m_deg_max      = hi;
        m_deg_avg      = sum / n;
        m_deg_sd       = std::sqrt(std::max(0.0, (sum2 - n * m_deg_avg * m_deg_avg) / (n > 1.0 ? n - 1.0 : 1.0)));

        std::vector<uint> sorted = deg;
        std::sort(sorted.begin(), sorted.end());
        const uint groups = std::max(1u, std::min<uint>(10u, hi - lo));
        m_deg_ranges.assign(1, lo);
        m_deg_distribution.clear();
        std::size_t from = 0;
        for (uint g = 1; g <= groups; ++g) {
            std::size_t to = g == groups ? sorted.size() : sorted.size() * g / groups;
            // keep equal degrees in one group
            while (to < sorted.size() && to > 0 && sorted[to] == sorted[to - 1]) ++to;
            if (to <= from) continue;
            m_deg_distribution.push_back(double(to - from) / n);
            m_deg_ranges.push_back(sorted[to - 1] + 1);
            from = to;
        }
    }

    void MtxLoader::output_stats() {
        std::cout << " deg: min " << m_deg_min << ", max " << m_deg_max << ", avg " << m_deg_avg << ", sd " << m_deg_sd << std::endl;
        std::cout << " share of vertices per degree range:";
        for (std::size_t g = 0; g < m_deg_distribution.size(); ++g)
            std::cout << " [" << m_deg_ranges[g] << "," << m_deg_ranges[g + 1] << ") " << m_deg_distribution[g];
        std::cout << std::endl;
    }

    const std::vector<uint>& MtxLoader::get_Ai() const { return m_Ai; }
    const std::vector<uint>& MtxLoader::get_Aj() const { return m_Aj; }
    uint                     MtxLoader::get_n_rows() const { return m_n_rows; }
    uint                     MtxLoader::get_n_cols() const { return m_n_cols; }
    std::size_t              MtxLoader::get_n_values() const { return m_n_values; }

}// namespace spla
