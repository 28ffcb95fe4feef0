// Native ZStore: content-addressed interning of Lurk data with LEVEL-ORDER batched hashing on the device.
//
// Replaces the hashing side of /root/reference/src/core/zstore.rs:
//   hash3 / hash4 / hash5 memo tables            zstore.rs:305-333   -> memo[width]
//   intern_tuple11 / intern_tuple110 + dag       zstore.rs:335-349   -> intern_dag (kinds 1, 2)
//   memoize_atom_dag                             zstore.rs:352-356   -> kind 0
//   intern_string / symbol / list / fun / env    zstore.rs:397-511   -> the caller flattens them into one node list (the
//                                                                        Python mirror lurk_amd/zstore.py does), intern_dag hashes it
//   memoize_dag                                  zstore.rs:569-702   -> memoize_dag (inverse tables, no hashing)
//   ZDag::populate_with_many                     cli/zdag.rs:16-55   -> dag_export
// The reference hashes one node at a time, children first (one Poseidon2 call per node).  Here a whole DAG of pending nodes
// arrives at once; nodes are grouped by height above the known data, every group is deduplicated against the memo tables
// and goes to the device as ONE lurkhip_poseidon2_hash8 launch per preimage width -- a syntax tree of depth d costs at most
// 3 d launches whatever its size.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/lurkhip.h"
#include "ctx.h"

namespace {

constexpr uint32_t P = 2013265921u;
// tag.rs:23-39
enum Tag : uint32_t { U64 = 0, Num, BigNum, Comm, Char, Str, Key, Fun, Builtin, Coroutine, Sym, Cons, Env, Fix, Err, N_TAGS };

using ZPtr = std::array<uint32_t, 9>;  // tag, digest[8]
struct ArrHash {
    template <size_t N>
    size_t operator()(const std::array<uint32_t, N>& a) const {
        uint64_t h = 0x9e3779b97f4a7c15ull ^ N;
        for (uint32_t x : a) {
            h = (h ^ x) * 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
        return (size_t)h;
    }
    size_t operator()(const std::vector<uint32_t>& a) const {
        uint64_t h = 0x9e3779b97f4a7c15ull ^ a.size();
        for (uint32_t x : a) {
            h = (h ^ x) * 0xff51afd7ed558ccdull;
            h ^= h >> 29;
        }
        return (size_t)h;
    }
};

struct ZType {
    uint32_t kind = 0;  // 0 Atom, 1 Tuple11, 2 Tuple110  (zstore.rs:213-219)
    ZPtr c[3] = {};
};

void flatten(const ZPtr& z, uint32_t* out) {  // zstore.rs:184-189: [tag, 0 x 7, digest]
    out[0] = z[0];
    for (int i = 1; i < 8; i++) out[i] = 0;
    memcpy(out + 8, &z[1], 32);
}

}  // namespace

struct lurkhip_zstore {
    lurkhip_ctx* ctx = nullptr;
    std::unordered_map<ZPtr, ZType, ArrHash> dag;
    std::unordered_map<std::vector<uint32_t>, std::array<uint32_t, 8>, ArrHash> memo;  // hashes3 / 4 / 5 (lengths differ)
    std::unordered_map<std::array<uint32_t, 8>, std::vector<uint32_t>, ArrHash> inv4, inv5;
    uint64_t hashed[3] = {0, 0, 0};  // permutations computed at widths 24 / 32 / 40
    uint64_t launches = 0, memo_hits = 0;
    std::string err;
};

namespace {

int32_t zfail(lurkhip_zstore* zs, int32_t code, const std::string& m) {
    zs->err = m;
    return lurkhip::set_error(zs->ctx, code, "%s", m.c_str());
}

// digests of `n` preimages of one width, through the memo table; the misses in one launch
int32_t hash_level(lurkhip_zstore* zs, int width, const std::vector<std::vector<uint32_t>>& preimgs, std::vector<std::array<uint32_t, 8>>& out) {
    out.resize(preimgs.size());
    std::vector<size_t> miss_first;                                       // first occurrence of every distinct missing preimage
    std::unordered_map<std::vector<uint32_t>, size_t, ArrHash> pending;   // preimage -> index into miss_first
    std::vector<long> slot(preimgs.size(), -1);
    for (size_t i = 0; i < preimgs.size(); i++) {
        auto it = zs->memo.find(preimgs[i]);
        if (it != zs->memo.end()) {
            out[i] = it->second;
            zs->memo_hits++;
            continue;
        }
        auto pit = pending.find(preimgs[i]);
        if (pit == pending.end()) {
            pit = pending.emplace(preimgs[i], miss_first.size()).first;
            miss_first.push_back(i);
        } else {
            zs->memo_hits++;
        }
        slot[i] = (long)pit->second;
    }
    if (miss_first.empty()) return LURKHIP_OK;
    std::vector<uint32_t> in(miss_first.size() * (size_t)width), dg(miss_first.size() * 8);
    for (size_t k = 0; k < miss_first.size(); k++) memcpy(&in[k * width], preimgs[miss_first[k]].data(), (size_t)width * 4);
    int32_t st = lurkhip_poseidon2_hash8(zs->ctx, width, miss_first.size(), in.data(), dg.data(), LURKHIP_REPR_CANONICAL);
    if (st != LURKHIP_OK) return st;
    zs->launches++;
    zs->hashed[width == 24 ? 0 : width == 32 ? 1 : 2] += miss_first.size();
    for (size_t k = 0; k < miss_first.size(); k++) {
        std::array<uint32_t, 8> d;
        memcpy(d.data(), &dg[k * 8], 32);
        zs->memo.emplace(preimgs[miss_first[k]], d);
    }
    for (size_t i = 0; i < preimgs.size(); i++)
        if (slot[i] >= 0) memcpy(out[i].data(), &dg[(size_t)slot[i] * 8], 32);
    return LURKHIP_OK;
}

}  // namespace

extern "C" {

int32_t lurkhip_zstore_new(lurkhip_ctx* ctx, lurkhip_zstore** out) {
    LH_CHECK_CTX(ctx);
    LH_ARG(ctx, out != nullptr, "null argument");
    auto* zs = new lurkhip_zstore();
    zs->ctx = ctx;
    *out = zs;
    return LURKHIP_OK;
}

int32_t lurkhip_zstore_free(lurkhip_zstore* zs) {
    delete zs;
    return LURKHIP_OK;
}

const char* lurkhip_zstore_last_error(const lurkhip_zstore* zs) { return zs ? zs->err.c_str() : ""; }

int32_t lurkhip_zstore_intern_dag(lurkhip_zstore* zs, uint32_t n_nodes, const uint32_t* nodes, uint32_t* out_zptrs) {
    if (!zs) return LURKHIP_ERR_INVALID_ARG;
    if (n_nodes && (!nodes || !out_zptrs)) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "null argument");
    try {
        std::vector<ZPtr> z(n_nodes);
        std::vector<uint32_t> level(n_nodes, 0);
        uint32_t max_level = 0;
        for (uint32_t i = 0; i < n_nodes; i++) {
            const uint32_t* nd = nodes + (size_t)i * LURKHIP_ZNODE_WORDS;
            const uint32_t kind = nd[0], tag = nd[1];
            if (tag >= N_TAGS) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "node " + std::to_string(i) + ": unknown tag");
            if (kind == LURKHIP_ZNODE_ATOM || kind == LURKHIP_ZNODE_REF) {
                z[i][0] = tag;
                for (int k = 0; k < 8; k++) {
                    if (nd[2 + k] >= P) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "node " + std::to_string(i) + ": digest lane is not canonical");
                    z[i][1 + k] = nd[2 + k];
                }
                if (kind == LURKHIP_ZNODE_ATOM) zs->dag.emplace(z[i], ZType{});  // memoize_atom_dag
                continue;
            }
            const int arity = kind == LURKHIP_ZNODE_TUPLE110 ? 3 : (kind == LURKHIP_ZNODE_TUPLE11 || kind == LURKHIP_ZNODE_COMM) ? 2 : 0;
            if (!arity) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "node " + std::to_string(i) + ": unknown kind");
            uint32_t lv = 0;
            for (int k = 0; k < arity; k++) {
                if (nd[2 + k] >= i) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "node " + std::to_string(i) + ": children must come before their parent");
                lv = std::max(lv, level[nd[2 + k]]);
            }
            level[i] = lv + 1;
            max_level = std::max(max_level, level[i]);
        }
        // nodes of one height form a batch per preimage width
        std::vector<std::vector<uint32_t>> by_level(max_level + 1);
        for (uint32_t i = 0; i < n_nodes; i++)
            if (level[i]) by_level[level[i]].push_back(i);
        for (uint32_t lv = 1; lv <= max_level; lv++) {
            std::vector<uint32_t> idx[3];
            std::vector<std::vector<uint32_t>> pre[3];
            for (uint32_t i : by_level[lv]) {
                const uint32_t* nd = nodes + (size_t)i * LURKHIP_ZNODE_WORDS;
                const uint32_t kind = nd[0];
                const ZPtr &a = z[nd[2]], &b = z[nd[3]];
                if (kind == LURKHIP_ZNODE_TUPLE11) {  // flatten_as_tuple11, zstore.rs:191-196
                    std::vector<uint32_t> p(32);
                    flatten(a, p.data());
                    flatten(b, p.data() + 16);
                    idx[1].push_back(i);
                    pre[1].push_back(std::move(p));
                } else if (kind == LURKHIP_ZNODE_TUPLE110) {  // flatten_as_tuple110, zstore.rs:198-204: the third child's tag is dropped
                    std::vector<uint32_t> p(40);
                    flatten(a, p.data());
                    flatten(b, p.data() + 16);
                    memcpy(p.data() + 32, &z[nd[4]][1], 32);
                    idx[2].push_back(i);
                    pre[2].push_back(std::move(p));
                } else {  // commitment: hash3(secret digest | flatten(payload))
                    std::vector<uint32_t> p(24);
                    memcpy(p.data(), &a[1], 32);
                    flatten(b, p.data() + 8);
                    idx[0].push_back(i);
                    pre[0].push_back(std::move(p));
                }
            }
            for (int wi = 0; wi < 3; wi++) {
                if (idx[wi].empty()) continue;
                std::vector<std::array<uint32_t, 8>> dg;
                int32_t st = hash_level(zs, wi == 0 ? 24 : wi == 1 ? 32 : 40, pre[wi], dg);
                if (st != LURKHIP_OK) {
                    zs->err = lurkhip_last_error(zs->ctx);
                    return st;
                }
                for (size_t k = 0; k < idx[wi].size(); k++) {
                    const uint32_t i = idx[wi][k];
                    const uint32_t* nd = nodes + (size_t)i * LURKHIP_ZNODE_WORDS;
                    z[i][0] = nd[0] == LURKHIP_ZNODE_COMM ? (uint32_t)Comm : nd[1];
                    memcpy(&z[i][1], dg[k].data(), 32);
                    ZType t;
                    if (nd[0] == LURKHIP_ZNODE_TUPLE11) {
                        t.kind = 1;
                        t.c[0] = z[nd[2]];
                        t.c[1] = z[nd[3]];
                    } else if (nd[0] == LURKHIP_ZNODE_TUPLE110) {
                        t.kind = 2;
                        t.c[0] = z[nd[2]];
                        t.c[1] = z[nd[3]];
                        t.c[2] = z[nd[4]];
                    }  // a commitment is an atom of the DAG (intern_comm, zstore.rs:388-391)
                    zs->dag[z[i]] = t;
                }
            }
        }
        for (uint32_t i = 0; i < n_nodes; i++) memcpy(out_zptrs + (size_t)i * 9, z[i].data(), 36);
        return LURKHIP_OK;
    } catch (const std::exception& e) {
        return zfail(zs, LURKHIP_ERR_EXEC, std::string("internal error: ") + e.what());
    }
}

int32_t lurkhip_zstore_stats(const lurkhip_zstore* zs, uint64_t* out) {
    if (!zs || !out) return LURKHIP_ERR_INVALID_ARG;
    out[0] = zs->hashed[0];
    out[1] = zs->hashed[1];
    out[2] = zs->hashed[2];
    out[3] = zs->launches;
    out[4] = zs->memo_hits;
    out[5] = zs->dag.size();
    return LURKHIP_OK;
}

int32_t lurkhip_zstore_set_inverse_tables(lurkhip_zstore* zs, uint64_t n4, const uint32_t* inv4, uint64_t n5, const uint32_t* inv5) {
    if (!zs || (n4 && !inv4) || (n5 && !inv5)) return LURKHIP_ERR_INVALID_ARG;
    try {
        zs->inv4.clear();
        zs->inv5.clear();
        for (uint64_t i = 0; i < n4; i++) {
            std::array<uint32_t, 8> d;
            memcpy(d.data(), inv4 + i * 40, 32);
            zs->inv4[d] = std::vector<uint32_t>(inv4 + i * 40 + 8, inv4 + i * 40 + 40);
        }
        for (uint64_t i = 0; i < n5; i++) {
            std::array<uint32_t, 8> d;
            memcpy(d.data(), inv5 + i * 48, 32);
            zs->inv5[d] = std::vector<uint32_t>(inv5 + i * 48 + 8, inv5 + i * 48 + 48);
        }
        return LURKHIP_OK;
    } catch (const std::exception& e) {
        return zfail(zs, LURKHIP_ERR_EXEC, std::string("internal error: ") + e.what());
    }
}

// zstore.rs:569-702, iteratively (the reference recurses on car / var / val and loops on cdr / env tails)
int32_t lurkhip_zstore_memoize_dag(lurkhip_zstore* zs, uint32_t tag, const uint32_t* digest) {
    if (!zs || !digest) return LURKHIP_ERR_INVALID_ARG;
    if (tag >= N_TAGS) return zfail(zs, LURKHIP_ERR_INVALID_ARG, "unknown tag");
    try {
        auto mk = [](uint32_t t, const uint32_t* d) {
            ZPtr z;
            z[0] = t;
            memcpy(&z[1], d, 32);
            return z;
        };
        const std::array<uint32_t, 8> zeros{};
        std::vector<ZPtr> work{mk(tag, digest)};
        while (!work.empty()) {
            ZPtr zp = work.back();
            work.pop_back();
            if (zs->dag.count(zp)) continue;
            std::array<uint32_t, 8> d;
            memcpy(d.data(), &zp[1], 32);
            switch (zp[0]) {
                case Str: {
                    if (d == zeros) {
                        zs->dag.emplace(zp, ZType{});
                        break;
                    }
                    auto it = zs->inv4.find(d);
                    if (it == zs->inv4.end()) return zfail(zs, LURKHIP_ERR_EXEC, "Hash4 preimg not found");
                    const uint32_t* p = it->second.data();
                    ZType t;
                    t.kind = 1;
                    t.c[0] = mk(Char, p + 8);   // the head's tag is taken to be Char, the tail's Str (zstore.rs:630-633)
                    t.c[1] = mk(Str, p + 24);
                    zs->dag.emplace(zp, t);
                    work.push_back(t.c[1]);
                    break;
                }
                case Cons: {
                    auto it = zs->inv4.find(d);
                    if (it == zs->inv4.end()) return zfail(zs, LURKHIP_ERR_EXEC, "Hash4 preimg not found");
                    const uint32_t* p = it->second.data();
                    if (p[0] >= N_TAGS || p[16] >= N_TAGS) return zfail(zs, LURKHIP_ERR_EXEC, "preimage carries an unknown tag");
                    ZType t;
                    t.kind = 1;
                    t.c[0] = mk(p[0], p + 8);
                    t.c[1] = mk(p[16], p + 24);
                    zs->dag.emplace(zp, t);
                    work.push_back(t.c[1]);
                    work.push_back(t.c[0]);
                    break;
                }
                case Env:
                case Fun:
                case Fix: {
                    if (zp[0] == Env && d == zeros) {
                        zs->dag.emplace(zp, ZType{});
                        break;
                    }
                    auto it = zs->inv5.find(d);
                    if (it == zs->inv5.end()) return zfail(zs, LURKHIP_ERR_EXEC, "Hash5 preimg not found");
                    const uint32_t* p = it->second.data();
                    if (p[0] >= N_TAGS || p[16] >= N_TAGS) return zfail(zs, LURKHIP_ERR_EXEC, "preimage carries an unknown tag");
                    ZType t;
                    t.kind = 2;
                    t.c[0] = mk(p[0], p + 8);
                    t.c[1] = mk(p[16], p + 24);
                    t.c[2] = mk(Env, p + 32);
                    zs->dag.emplace(zp, t);
                    work.push_back(t.c[2]);
                    work.push_back(t.c[1]);
                    work.push_back(t.c[0]);
                    break;
                }
                case Sym:
                case Key:
                case Builtin:
                case Coroutine:
                    break;  // "these should be already memoized" (zstore.rs:692)
                default:  // Num, U64, Char, Err, BigNum, Comm
                    zs->dag.emplace(zp, ZType{});
            }
        }
        return LURKHIP_OK;
    } catch (const std::exception& e) {
        return zfail(zs, LURKHIP_ERR_EXEC, std::string("internal error: ") + e.what());
    }
}

// out[0] = kind (0 Atom, 1 Tuple11, 2 Tuple110), out[1 .. 28) = the children (9 words each, unused ones zero)
int32_t lurkhip_zstore_fetch(const lurkhip_zstore* zs, const uint32_t* zptr, uint32_t* out) {
    if (!zs || !zptr || !out) return LURKHIP_ERR_INVALID_ARG;
    ZPtr z;
    memcpy(z.data(), zptr, 36);
    auto it = zs->dag.find(z);
    if (it == zs->dag.end()) return LURKHIP_ERR_INVALID_ARG;  // "Data missing from ZStore's DAG"
    out[0] = it->second.kind;
    for (int k = 0; k < 3; k++) memcpy(out + 1 + 9 * k, it->second.c[k].data(), 36);
    return LURKHIP_OK;
}

// ZDag::populate_with_many (cli/zdag.rs:16-55): entries reachable from the roots, children before parents, each once.
// Entry = 37 words: zptr (9), kind (1), children (27).  Returns the number of entries (writes at most cap_entries).
int64_t lurkhip_zstore_dag_export(const lurkhip_zstore* zs, uint32_t n_roots, const uint32_t* roots, uint32_t* out, uint64_t cap_entries) {
    if (!zs || (n_roots && !roots)) return LURKHIP_ERR_INVALID_ARG;
    try {
        std::unordered_map<ZPtr, bool, ArrHash> seen;
        uint64_t n = 0;
        struct Frame {
            ZPtr z;
            int next;
        };
        for (uint32_t r = 0; r < n_roots; r++) {
            ZPtr root;
            memcpy(root.data(), roots + (size_t)r * 9, 36);
            std::vector<Frame> stack{{root, 0}};
            while (!stack.empty()) {
                Frame& f = stack.back();
                if (f.next == 0 && seen.count(f.z)) {
                    stack.pop_back();
                    continue;
                }
                auto it = zs->dag.find(f.z);
                if (it == zs->dag.end()) return LURKHIP_ERR_EXEC;  // "Data missing from ZStore's DAG"
                const int arity = it->second.kind == 0 ? 0 : it->second.kind == 1 ? 2 : 3;
                if (f.next < arity) {
                    ZPtr child = it->second.c[f.next++];
                    stack.push_back({child, 0});
                    continue;
                }
                seen[f.z] = true;
                if (out && n < cap_entries) {
                    uint32_t* e = out + n * 37;
                    memcpy(e, f.z.data(), 36);
                    e[9] = it->second.kind;
                    for (int k = 0; k < 3; k++) memcpy(e + 10 + 9 * k, it->second.c[k].data(), 36);
                }
                n++;
                stack.pop_back();
            }
        }
        return (int64_t)n;
    } catch (const std::exception&) {
        return LURKHIP_ERR_EXEC;
    }
}

}  // extern "C"
