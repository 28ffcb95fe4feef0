// Index arithmetic of a commitment made by G = 2^log_g ranks together (one shard over several GPUs; SURVEY.md 8e, bullet 2;
// collective C2 of SURVEY.md 7).  Host only, no HIP: split.hip runs the plan on device buffers, lurkhip_split_plan hands it to a
// test that runs it on host arrays over gloo.
//
// The reference proves a shard inside ONE process: `Shard::shard` only cuts an execution above 2^22 rows
// (/root/reference/src/lair/execute.rs:186-241), so `machine.prove` (/root/reference/benches/fib.rs:124) of anything smaller is one
// shard however many GPUs there are.  Here the matrices of such a shard are cut two ways:
//   * the coset LDE couples all rows of a column, so it runs on COLUMN tiles: the matrices of one height are one virtual row of
//     W columns, rank r transforms columns [bounds[r], bounds[r + 1]);
//   * leaf hashing, the quotient, the reduced openings and the Merkle paths are row-local on the bit-reversed LDE, so afterwards
//     every rank holds the contiguous STORAGE rows [r * 2N / G, (r + 1) * 2N / G) of every matrix: a subtree of the Merkle tree;
//   * one all-to-all in between (exchange B); and one before the LDE (exchange A) when the source rows were computed by row blocks
//     (permutation traces, quotient values) rather than replicated (main traces in the first version, preprocessed traces).
// Matrices below 2^min_log_n rows are not cut: every rank computes them whole ("small").
#pragma once
#include <stdint.h>

#include <algorithm>
#include <stdexcept>
#include <vector>

namespace lurkhip {
namespace split {

enum Kind : int {
    K_FULL = 0,      // every rank holds all N rows (natural order)
    K_BLOCK = 1,     // rank r holds natural rows [r N / G, (r + 1) N / G)
    K_QUOTIENT = 2,  // chunk `chunk` of a chip's quotient values, computed on the rank's storage rows of the quotient domain
};

struct MatDesc {
    uint32_t log_n, width;
    int kind;
    uint32_t lqd, chunk;  // K_QUOTIENT
    uint32_t n_next;      // next-row copies of columns 0 .. n_next - 1 travel with exchange B (main traces: lair::ChipAir's main_next columns)
    uint32_t next_lqd;    // ... on the quotient domain 2^(log_n + next_lqd)
    // the ascending, disjoint (first column, width) runs outside which the matrix is identically zero on EVERY rank (the permutation
    // traces' dead batch columns, agreed by all ranks): only these columns are transformed and exchanged, the others stay the zeros
    // the rank's row block is filled with.  Empty: every column.
    std::vector<std::pair<uint32_t, uint32_t>> runs;
};

struct RowSet {
    uint32_t start, stride, count;  // natural rows start + k * stride, k < count (count 0: none)
};

inline uint32_t brev_bits(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

// The rows of matrix m that rank `rank` holds before exchange A.
// K_QUOTIENT: the quotient domain g <w_Q>, Q = N << lqd, is the first Q storage rows of the LDE (bit-reversed order); rank r holds
// LDE storage rows [r L2, (r + 1) L2), L2 = 2N / G.  Storage row s = r L2 + j is natural index i = brev_logQ(s)
// = brev(j) * 2^t + brev_t(r) with t = log Q - log L2 = lqd - 1 + log_g; it belongs to chunk i mod 2^lqd, row i >> lqd of that chunk:
// row = brev(j) * 2^(t - lqd) + (brev_t(r) >> lqd).  The quotient kernel writes its local values in brev(j) order, so the rank's rows of
// its chunk are an arithmetic progression.
inline RowSet rows_of(const MatDesc& m, int rank, int log_g) {
    const uint32_t n = 1u << m.log_n;
    if (m.kind == K_FULL) return {0, 1, n};
    if (m.kind == K_BLOCK) return {(uint32_t)rank * (n >> log_g), 1, n >> log_g};
    const int t = (int)m.lqd - 1 + log_g;
    if (t < (int)m.lqd) throw std::runtime_error("split: a quotient matrix needs at least two ranks");
    if ((uint32_t)rank >= (1u << t)) return {0, 1, 0};  // the rank's storage rows lie outside the quotient domain
    const uint32_t rb = brev_bits((uint32_t)rank, t);
    if ((rb & ((1u << m.lqd) - 1u)) != m.chunk) return {0, 1, 0};
    return {rb >> m.lqd, 1u << (t - (int)m.lqd), (n << 1) >> log_g};
}

struct Extra {
    int mat;
    uint32_t col;    // column of the matrix whose next-row copy this is
    uint32_t vcol;   // its column in the group's virtual row
    int owner;       // the rank whose tile holds vcol
};

struct Segment {  // a live run of a matrix: columns [c0, c0 + w) of it = columns [vstart, vstart + w) of the group's virtual row
    int mat;
    uint32_t c0, w, vstart;
};

struct Group {
    uint32_t log_n = 0;
    std::vector<int> mats;            // committed order
    std::vector<uint32_t> col_start;  // first column of mats[k] in the rank's row block (every column of every matrix, dead ones too)
    uint32_t W_local = 0;             // the matrices' widths side by side: the row block's columns (next-row copies follow them)
    std::vector<Segment> segs;        // the virtual row the LDE and the exchanges see: live columns only
    uint32_t W = 0;                   // its width
    std::vector<uint32_t> bounds;     // G + 1 column bounds
    std::vector<Extra> extras;
    uint32_t local_pitch = 0;         // words between the rows of the rank's row block [2N / G][W_local + extras], a multiple of 32
    bool sparse = false;              // some column of some matrix is left out: the row block is zero-filled before exchange B lands in it
};

// One strided copy between a matrix-shaped buffer and a linear (send / receive) buffer.
struct Job {
    int buf;                         // which matrix-side buffer (meaning depends on the list, see Plan)
    uint32_t row0, col0, row_stride; // matrix side: rows row0 + k * row_stride, columns col0 ..
    uint64_t lin_off;                // linear side: word offset of the block
    uint32_t lin_pitch;              // words between its rows
    uint32_t width, rows;
};

struct Tile {      // columns [c0, c0 + w) of matrix `mat`: one LDE input of this rank
    int group, mat;
    uint32_t c0, w;
    uint32_t slab_col;  // for exchanged sources: its first column in the rank's slab of the group
};

struct Plan {
    int log_g = 0, rank = 0, min_log_n = 0;
    std::vector<Group> groups;   // split heights, tallest first
    std::vector<int> group_of;   // per matrix: group index, -1: small (not cut)
    std::vector<Tile> tiles;     // this rank's LDE inputs: group order, then virtual column order
    std::vector<int> my_extras;  // (group << 16 | index into groups[g].extras) of the next-row copies this rank makes
    std::vector<uint32_t> slab_w;  // per group: this rank's tile width = the pitch of its slab [N][slab_w]
    // exchange A: rows -> column tiles.  a_pack: buf = matrix index (the rank's local rows of it); a_unpack: buf = group (slab)
    bool has_a = false;
    std::vector<uint64_t> a_send_off, a_recv_off;  // G + 1 word offsets
    std::vector<Job> a_pack, a_unpack;
    // exchange B: column tiles of the LDE -> row blocks.  b_pack: buf = tile index, or tiles.size() + k for my_extras[k];
    // b_unpack: buf = group (the rank's row block)
    std::vector<uint64_t> b_send_off, b_recv_off;
    std::vector<Job> b_pack, b_unpack;
};

// The LDE kernels transform 32-column tiles: a rank pays a whole tile pass for any share of 32 columns.
constexpr uint32_t LDE_TILE_COLS = 32;
inline uint32_t tiles_of(uint32_t cols) { return (cols + LDE_TILE_COLS - 1) / LDE_TILE_COLS; }

// `load` (G entries, updated): rows x tiles every rank has been given so far, by the groups before this one.
inline std::vector<uint32_t> column_bounds(uint32_t W, int G, uint32_t log_n, std::vector<uint64_t>& load) {
    std::vector<uint32_t> b((size_t)G + 1);
    if (W < LDE_TILE_COLS * (uint32_t)G) {
        // A NARROW group -- fewer than one tile a rank (a height's quotient chunks, the short chips): an even share would be a few
        // columns on EVERY rank, each of them paying the whole tile pass, group after group.  Whole tiles instead, to as many
        // consecutive ranks as there are tiles, starting where the ranks have the least to do so far: the groups of a commitment
        // spread over the ranks and run side by side.
        const uint32_t units = tiles_of(W);
        uint32_t best = 0;
        uint64_t best_max = ~(uint64_t)0;
        for (uint32_t start = 0; start + units <= (uint32_t)G; start++) {
            uint64_t m = 0;
            for (uint32_t k = 0; k < units; k++) m = std::max(m, load[(size_t)(start + k)]);
            if (m < best_max) {
                best_max = m;
                best = start;
            }
        }
        for (int r = 0; r <= G; r++) {
            const uint32_t before = (uint32_t)r <= best ? 0u : std::min((uint32_t)r - best, units);  // tiles given to the ranks below r
            b[(size_t)r] = std::min(W, before * LDE_TILE_COLS);
        }
    } else {
        // whole 32-column tiles while every rank still gets two of them, else fours
        const uint32_t unit = W >= 2 * LDE_TILE_COLS * (uint32_t)G ? LDE_TILE_COLS : 4u;
        const uint32_t units = W / unit;
        for (int r = 0; r < G; r++) b[(size_t)r] = unit * (uint32_t)(((uint64_t)units * (uint64_t)r) / (uint64_t)G);
        b[(size_t)G] = W;
    }
    for (int r = 0; r < G; r++) load[(size_t)r] += (uint64_t)tiles_of(b[(size_t)r + 1] - b[(size_t)r]) << log_n;
    return b;
}

inline Plan make_plan(int log_g, int rank, int min_log_n, const std::vector<MatDesc>& mats) {
    if (log_g < 1 || log_g > 6) throw std::runtime_error("split: 2 to 64 ranks");
    if (min_log_n < log_g) throw std::runtime_error("split: min_log_n below log2(ranks)");
    const int G = 1 << log_g;
    if (rank < 0 || rank >= G) throw std::runtime_error("split: bad rank");
    Plan p;
    p.log_g = log_g;
    p.rank = rank;
    p.min_log_n = min_log_n;
    p.group_of.assign(mats.size(), -1);
    // groups: split heights, tallest first, members in committed order
    std::vector<uint32_t> heights;
    for (const MatDesc& m : mats)
        if ((int)m.log_n >= min_log_n && std::find(heights.begin(), heights.end(), m.log_n) == heights.end()) heights.push_back(m.log_n);
    std::sort(heights.rbegin(), heights.rend());
    std::vector<uint64_t> load((size_t)G, 0);
    for (uint32_t h : heights) {
        Group g;
        g.log_n = h;
        for (size_t i = 0; i < mats.size(); i++)
            if (mats[i].log_n == h) {
                const MatDesc& m = mats[i];
                p.group_of[i] = (int)p.groups.size();
                g.mats.push_back((int)i);
                g.col_start.push_back(g.W_local);
                g.W_local += m.width;
                if (m.runs.empty()) {
                    g.segs.push_back(Segment{(int)i, 0, m.width, g.W});
                    g.W += m.width;
                } else {
                    uint32_t at = 0;
                    for (const auto& r : m.runs) {
                        if (r.first < at || r.second == 0 || r.first + r.second > m.width) throw std::runtime_error("split: malformed live runs");
                        if (m.n_next) throw std::runtime_error("split: live runs and next-row copies on one matrix");
                        g.segs.push_back(Segment{(int)i, r.first, r.second, g.W});
                        g.W += r.second;
                        at = r.first + r.second;
                    }
                    uint32_t covered = 0;
                    for (const auto& r : m.runs) covered += r.second;
                    g.sparse = g.sparse || covered < m.width;
                }
            }
        g.bounds = column_bounds(g.W, G, g.log_n, load);
        for (size_t k = 0; k < g.mats.size(); k++) {
            const MatDesc& m = mats[(size_t)g.mats[k]];
            if (m.n_next > m.width) throw std::runtime_error("split: more next-row columns than columns");
            if (m.n_next == 0) continue;
            uint32_t vstart = 0;  // (a matrix with next-row copies has one segment: all of it)
            for (const Segment& sg : g.segs)
                if (sg.mat == g.mats[k]) vstart = sg.vstart;
            for (uint32_t c = 0; c < m.n_next; c++) {
                Extra e{g.mats[k], c, vstart + c, 0};
                while (g.bounds[(size_t)e.owner + 1] <= e.vcol) e.owner++;
                g.extras.push_back(e);
            }
        }
        g.local_pitch = (g.W_local + (uint32_t)g.extras.size() + 31u) & ~31u;
        p.groups.push_back(g);
    }
    // this rank's tiles
    p.slab_w.assign(p.groups.size(), 0);
    for (size_t gi = 0; gi < p.groups.size(); gi++) {
        const Group& g = p.groups[gi];
        const uint32_t lo = g.bounds[(size_t)rank], hi = g.bounds[(size_t)rank + 1];
        p.slab_w[gi] = hi - lo;
        for (const Segment& sg : g.segs) {
            const uint32_t a = std::max(lo, sg.vstart), b = std::min(hi, sg.vstart + sg.w);
            if (a < b) p.tiles.push_back(Tile{(int)gi, sg.mat, sg.c0 + (a - sg.vstart), b - a, a - lo});
        }
        for (size_t e = 0; e < g.extras.size(); e++)
            if (g.extras[e].owner == rank) p.my_extras.push_back((int)((gi << 16) | e));
    }
    // ---- exchange A.  Payload of sender s for destination d: for every group, for every matrix of the group that is exchanged
    // (kind != K_FULL), of which s holds rows and whose columns meet d's tile: the block [rows of s][columns in d's tile].
    auto a_walk = [&](int s, int d, auto&& emit) {
        uint64_t at = 0;
        for (size_t gi = 0; gi < p.groups.size(); gi++) {
            const Group& g = p.groups[gi];
            const uint32_t lo = g.bounds[(size_t)d], hi = g.bounds[(size_t)d + 1];
            for (const Segment& sg : g.segs) {
                const MatDesc& m = mats[(size_t)sg.mat];
                if (m.kind == K_FULL) continue;
                const RowSet rs = rows_of(m, s, log_g);
                const uint32_t a = std::max(lo, sg.vstart), b = std::min(hi, sg.vstart + sg.w);
                if (rs.count == 0 || a >= b) continue;
                emit(gi, sg.mat, rs, sg.c0 + (a - sg.vstart), b - a, a - lo, at);
                at += (uint64_t)rs.count * (b - a);
            }
        }
        return at;
    };
    for (const MatDesc& m : mats)
        if ((int)m.log_n >= min_log_n && m.kind != K_FULL) p.has_a = true;
    if (p.has_a) {
        p.a_send_off.assign((size_t)G + 1, 0);
        p.a_recv_off.assign((size_t)G + 1, 0);
        for (int d = 0; d < G; d++) {
            const uint64_t base = p.a_send_off[(size_t)d];
            const uint64_t words = a_walk(rank, d, [&](size_t, int mat, const RowSet& rs, uint32_t c0, uint32_t w, uint32_t, uint64_t at) {
                p.a_pack.push_back(Job{mat, 0, c0, 1, base + at, w, w, rs.count});
            });
            p.a_send_off[(size_t)d + 1] = base + words;
        }
        for (int s = 0; s < G; s++) {
            const uint64_t base = p.a_recv_off[(size_t)s];
            const uint64_t words = a_walk(s, rank, [&](size_t gi, int, const RowSet& rs, uint32_t, uint32_t w, uint32_t slab_col, uint64_t at) {
                p.a_unpack.push_back(Job{(int)gi, rs.start, slab_col, rs.stride, base + at, w, w, rs.count});
            });
            p.a_recv_off[(size_t)s + 1] = base + words;
        }
    }
    // ---- exchange B.  Payload of sender s for destination d: for every group the block [L2][tile width of s + extras s makes],
    // rows = d's storage rows of the LDE, columns = s's tile in virtual order, then its next-row copies in the group's extras order.
    auto wsend = [&](int s, const Group& g) {
        uint32_t w = g.bounds[(size_t)s + 1] - g.bounds[(size_t)s];
        for (const Extra& e : g.extras) w += e.owner == s ? 1u : 0u;
        return w;
    };
    p.b_send_off.assign((size_t)G + 1, 0);
    p.b_recv_off.assign((size_t)G + 1, 0);
    for (int d = 0; d < G; d++) {
        uint64_t at = p.b_send_off[(size_t)d];
        size_t tile = 0, ex = 0;
        for (size_t gi = 0; gi < p.groups.size(); gi++) {
            const Group& g = p.groups[gi];
            const uint32_t l2 = (2u << g.log_n) >> log_g, ws = wsend(rank, g);
            for (; tile < p.tiles.size() && p.tiles[tile].group == (int)gi; tile++)
                p.b_pack.push_back(Job{(int)tile, (uint32_t)d * l2, 0, 1, at + p.tiles[tile].slab_col, ws, p.tiles[tile].w, l2});
            uint32_t col = g.bounds[(size_t)rank + 1] - g.bounds[(size_t)rank];
            for (; ex < p.my_extras.size() && (size_t)(p.my_extras[ex] >> 16) == gi; ex++, col++)
                p.b_pack.push_back(Job{(int)(p.tiles.size() + ex), (uint32_t)d * l2, 0, 1, at + col, ws, 1, l2});
            at += (uint64_t)l2 * ws;
        }
        p.b_send_off[(size_t)d + 1] = at;
    }
    for (int s = 0; s < G; s++) {
        uint64_t at = p.b_recv_off[(size_t)s];
        for (size_t gi = 0; gi < p.groups.size(); gi++) {
            const Group& g = p.groups[gi];
            const uint32_t l2 = (2u << g.log_n) >> log_g, ws = wsend(s, g), tw = g.bounds[(size_t)s + 1] - g.bounds[(size_t)s];
            // s's tile, segment by segment: virtual columns [lo, hi) land at their matrices' columns of the row block
            const uint32_t lo = g.bounds[(size_t)s], hi = g.bounds[(size_t)s + 1];
            for (const Segment& sg : g.segs) {
                const uint32_t a = std::max(lo, sg.vstart), b = std::min(hi, sg.vstart + sg.w);
                if (a >= b) continue;
                size_t k = 0;
                while (g.mats[k] != sg.mat) k++;
                p.b_unpack.push_back(Job{(int)gi, 0, g.col_start[k] + sg.c0 + (a - sg.vstart), 1, at + (a - lo), ws, b - a, l2});
            }
            uint32_t col = tw;
            for (size_t e = 0; e < g.extras.size(); e++)
                if (g.extras[e].owner == s) p.b_unpack.push_back(Job{(int)gi, 0, g.W_local + (uint32_t)e, 1, at + col++, ws, 1, l2});
            at += (uint64_t)l2 * ws;
        }
        p.b_recv_off[(size_t)s + 1] = at;
    }
    return p;
}

}  // namespace split
}  // namespace lurkhip
