// Property check of lurk_amd/csrc/split_plan.h (host only, no HIP) under AddressSanitizer + UBSan: random commitments -- ragged
// widths, the three kinds of row sources, next-row copies, dead column runs -- for G = 2 .. 64 ranks.  For every shape the G plans
// must agree with each other: what s sends to d is what d expects from s, block by block; every job stays inside its buffers;
// every word of every rank's row block that should arrive does arrive exactly once.
// Built and run by tests/test_split_plan_sanitized.py (g++ -fsanitize=address,undefined).
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../lurk_amd/csrc/split_plan.h"

using namespace lurkhip::split;

static int fail(const char* what, int G, int seed) {
    std::fprintf(stderr, "FAIL: %s (G = %d, seed %d)\n", what, G, seed);
    return 1;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 40;
    long checked = 0;
    for (int seed = 0; seed < rounds; seed++) {
        std::mt19937 rng((unsigned)seed * 7919u + 13u);
        for (int log_g = 1; log_g <= 6; log_g++) {
            const int G = 1 << log_g;
            const int min_log_n = log_g + (int)(rng() % 3);
            std::vector<MatDesc> mats;
            const int n_mats = 1 + (int)(rng() % 9);
            for (int i = 0; i < n_mats; i++) {
                MatDesc m{};
                m.log_n = (uint32_t)(rng() % 3 == 0 ? rng() % (unsigned)(min_log_n + 1) : min_log_n + rng() % 4);
                m.width = 1 + (uint32_t)(rng() % (rng() % 4 == 0 ? 700 : 40));
                const bool tall = (int)m.log_n >= min_log_n;
                m.kind = tall ? (int)(rng() % 3) : K_FULL;
                if (m.kind == K_QUOTIENT) {
                    m.lqd = (uint32_t)(rng() % 2);
                    m.chunk = m.lqd ? (uint32_t)(rng() % 2) : 0u;
                    m.width = 4;
                }
                m.next_lqd = 1;
                if (m.kind == K_FULL && rng() % 3 == 0) m.n_next = 1 + (uint32_t)(rng() % std::min<uint32_t>(m.width, 3u));
                if (m.kind == K_BLOCK && m.n_next == 0 && rng() % 2 == 0) {  // dead columns
                    uint32_t at = 0;
                    while (at < m.width) {
                        const uint32_t skip = (uint32_t)(rng() % 5), w = 1 + (uint32_t)(rng() % 9);
                        at += skip;
                        if (at >= m.width) break;
                        const uint32_t ww = std::min(w, m.width - at);
                        m.runs.push_back({at, ww});
                        at += ww;
                    }
                    if (m.runs.empty()) m.runs.push_back({m.width - 1, 1u});
                }
                mats.push_back(m);
            }
            std::vector<Plan> plans;
            try {
                for (int r = 0; r < G; r++) plans.push_back(make_plan(log_g, r, min_log_n, mats));
            } catch (const std::exception& e) {
                std::fprintf(stderr, "%s\n", e.what());
                return fail("make_plan threw", G, seed);
            }
            const Plan& p0 = plans[0];
            for (size_t gi = 0; gi < p0.groups.size(); gi++) {
                const Group& g = p0.groups[gi];
                if (g.bounds.front() != 0 || g.bounds.back() != g.W) return fail("bounds do not cover the virtual row", G, seed);
                for (int r = 0; r < G; r++)
                    if (g.bounds[(size_t)r] > g.bounds[(size_t)r + 1]) return fail("bounds not monotone", G, seed);
                if (g.local_pitch % 32 || g.local_pitch < g.W_local + g.extras.size()) return fail("row block pitch", G, seed);
            }
            for (int s = 0; s < G; s++)
                for (int d = 0; d < G; d++) {
                    if (p0.has_a && plans[(size_t)s].a_send_off[(size_t)d + 1] - plans[(size_t)s].a_send_off[(size_t)d] !=
                                        plans[(size_t)d].a_recv_off[(size_t)s + 1] - plans[(size_t)d].a_recv_off[(size_t)s])
                        return fail("exchange A: sender and receiver disagree on a block's size", G, seed);
                    if (plans[(size_t)s].b_send_off[(size_t)d + 1] - plans[(size_t)s].b_send_off[(size_t)d] !=
                        plans[(size_t)d].b_recv_off[(size_t)s + 1] - plans[(size_t)d].b_recv_off[(size_t)s])
                        return fail("exchange B: sender and receiver disagree on a block's size", G, seed);
                }
            for (int r = 0; r < G; r++) {
                const Plan& p = plans[(size_t)r];
                // exchange B's unpack: every live column (and next-row copy) of every row of the rank's blocks is written exactly once
                std::vector<std::vector<uint8_t>> hit(p.groups.size());
                for (size_t gi = 0; gi < p.groups.size(); gi++) hit[gi].assign((size_t)p.groups[gi].local_pitch, 0);
                for (const Job& j : p.b_unpack) {
                    const Group& g = p.groups[(size_t)j.buf];
                    const uint32_t l2 = (2u << g.log_n) >> log_g;
                    if (j.rows != l2 || j.row0 != 0 || j.row_stride != 1) return fail("exchange B: an unpack job does not cover the block's rows", G, seed);
                    if (j.col0 + j.width > g.local_pitch) return fail("exchange B: an unpack job leaves the row block", G, seed);
                    if (j.lin_off + (uint64_t)(j.rows - 1) * j.lin_pitch + j.width > p.b_recv_off.back()) return fail("exchange B: an unpack job leaves the receive buffer", G, seed);
                    for (uint32_t c = 0; c < j.width; c++)
                        if (hit[(size_t)j.buf][(size_t)j.col0 + c]++) return fail("exchange B: a column is written twice", G, seed);
                }
                for (size_t gi = 0; gi < p.groups.size(); gi++) {
                    const Group& g = p.groups[gi];
                    std::vector<uint8_t> want((size_t)g.local_pitch, 0);
                    for (const Segment& sg : g.segs) {
                        size_t k = 0;
                        while (g.mats[k] != sg.mat) k++;
                        for (uint32_t c = 0; c < sg.w; c++) want[(size_t)g.col_start[k] + sg.c0 + c] = 1;
                    }
                    for (size_t e = 0; e < g.extras.size(); e++) want[(size_t)g.W_local + e] = 1;
                    if (want != hit[gi]) return fail("exchange B: the columns written are not the live columns", G, seed);
                }
                for (const Job& j : p.b_pack) {
                    if (j.lin_off + (uint64_t)(j.rows - 1) * j.lin_pitch + j.width > p.b_send_off.back()) return fail("exchange B: a pack job leaves the send buffer", G, seed);
                    if ((size_t)j.buf >= p.tiles.size() + p.my_extras.size()) return fail("exchange B: a pack job of no tile", G, seed);
                }
                // exchange A's unpack: every row and column of the rank's slabs that an exchanged matrix fills is written exactly once
                if (p.has_a) {
                    for (const Job& j : p.a_pack)
                        if (j.lin_off + (uint64_t)(j.rows - 1) * j.lin_pitch + j.width > p.a_send_off.back()) return fail("exchange A: a pack job leaves the send buffer", G, seed);
                    std::vector<std::vector<uint32_t>> cells(p.groups.size());
                    for (size_t gi = 0; gi < p.groups.size(); gi++) cells[gi].assign(((size_t)1 << p.groups[gi].log_n) * std::max<uint32_t>(p.slab_w[gi], 1), 0);
                    for (const Job& j : p.a_unpack) {
                        const Group& g = p.groups[(size_t)j.buf];
                        const uint32_t sw = p.slab_w[(size_t)j.buf];
                        if (j.col0 + j.width > sw) return fail("exchange A: an unpack job leaves the slab", G, seed);
                        if (j.lin_off + (uint64_t)(j.rows - 1) * j.lin_pitch + j.width > p.a_recv_off.back()) return fail("exchange A: an unpack job leaves the receive buffer", G, seed);
                        for (uint32_t k = 0; k < j.rows; k++) {
                            const uint64_t row = (uint64_t)j.row0 + (uint64_t)k * j.row_stride;
                            if (row >> g.log_n) return fail("exchange A: an unpack job writes past the last row", G, seed);
                            for (uint32_t c = 0; c < j.width; c++)
                                if (cells[(size_t)j.buf][row * sw + j.col0 + c]++) return fail("exchange A: a cell is written twice", G, seed);
                        }
                    }
                    for (const Tile& t : p.tiles) {
                        if (mats[(size_t)t.mat].kind == K_FULL) continue;
                        const Group& g = p.groups[(size_t)t.group];
                        const uint32_t sw = p.slab_w[(size_t)t.group];
                        for (uint64_t row = 0; row < ((uint64_t)1 << g.log_n); row++)
                            for (uint32_t c = 0; c < t.w; c++)
                                if (cells[(size_t)t.group][row * sw + t.slab_col + c] != 1) return fail("exchange A: a cell of a tile is not filled", G, seed);
                    }
                }
                checked++;
            }
        }
    }
    std::printf("ok: %ld plans\n", checked);
    return 0;
}
