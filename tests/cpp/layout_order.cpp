// layout_order.cpp -- part of libmipt_diag.so (test infrastructure): the HOST restatement of the two orders of the device layout --
// where each 64-byte pair record and each triangle's 64-byte intersection record sits in HBM.  The product computes both with GPU
// kernels (rust_ray_tracing_amd/csrc/scene_device.hip: DoublesOp / RestOp / LevelOp); these functions, which laid the scene out on the
// host in rounds 1-3, are what tests/cpp/host_layout.cpp, tests/test_host_layout.py and tests/tools/layout_model.py check them against.
#include "../../include/mipt.h"
#include "../../rust_ray_tracing_amd/csrc/mipt_internal.h"

#include <cstring>
#include <exception>
#include <vector>

// ---- order of the device's 64-B pair records (pair k = {nodes[2k+1], nodes[2k+2]}) ----
// The memory side moves whole 128-B lines and a traversal step gathers ONE 64-B record, so what matters is which record shares a
// record's line.  Two zones:
//   * the top kPairLayoutTop levels, which every ray walks through and which stay resident in L1 / L2: breadth-first, level after
//     level, every level starting on a line boundary -- one dense run of lines;
//   * below them, where a line is cold whenever a ray reaches it: a pair shares its line with the child pair of its LARGER inner
//     child (half area: SAH's own proxy for "the child the ray enters").  The step after a cold pair is then a hit more often than
//     not.  A pair that was taken into its parent's line leaves its own children to head new lines; pairs with two leaf children
//     that nobody took are packed two by two at the end, in level order (siblings are neighbours there).
// Measured on config M (tools/ab_pmc.sh, profiles/r3_layout_ab.csv): 12.8 -> 10.0 line fills per ray against the round-2 order
// (breadth-first with the two child pairs of a node in one line, which only moved hits from L2 to L1); the order was picked with
// the replay model of tests/tools/layout_model.py (predicted 12.2 -> 10.0).  Topology, visit order and results are untouched.
// order_out[j] = reference pair index of record j, or 0xffffffff for a pad record; *n_records_out = number of records.
using mipt::kPairLayoutTop;
int mipt::pair_order(const MiptNode *nodes, uint32_t n_nodes, uint32_t *order_out, uint32_t cap, uint32_t *n_records_out) {
    if (!nodes || !order_out || !n_records_out || (n_nodes & 1u) == 0u) return MIPT_ERR_INVALID_ARG;
    try {
        const uint32_t n_pairs = (n_nodes - 1u) / 2u;
        std::vector<uint32_t> order;
        order.reserve((size_t)n_pairs + 64);
        auto child_pair = [&](uint32_t k, uint32_t w, uint32_t *out) -> bool {
            const MiptNode &n = nodes[2 * k + 1 + w];
            if (n.num_tris != 0u) return false;
            *out = (n.first_tri_or_child - 1u) / 2u;
            return true;
        };
        std::vector<uint32_t> level, couples, singles;
        if (n_pairs > 0) level.push_back(0u);
        std::vector<uint8_t> taken(n_pairs, 0);
        std::vector<uint32_t> lone;
        auto half_area = [&](uint32_t node) -> double {
            const MiptNode &n = nodes[node];
            const double ex = (double)n.bounds_max.x - n.bounds_min.x, ey = (double)n.bounds_max.y - n.bounds_min.y, ez = (double)n.bounds_max.z - n.bounds_min.z;
            return ex * ey + ey * ez + ez * ex;
        };
        uint32_t depth = 0;
        while (!level.empty()) {
            couples.clear(); singles.clear();
            if (depth < kPairLayoutTop) {
                if (order.size() + level.size() + 1 > (size_t)cap) return MIPT_ERR_SCENE_LIMIT;   // also bounds a malformed (shared-child) input
                if (order.size() & 1u) order.push_back(0xffffffffu);
                for (uint32_t k : level) order.push_back(k);
            } else {
                for (uint32_t k : level) {
                    if (taken[k]) continue;
                    uint32_t ca = 0, cb = 0;
                    const bool ha = child_pair(k, 0, &ca), hb = child_pair(k, 1, &cb);
                    if (!ha && !hb) { lone.push_back(k); continue; }
                    uint32_t pick = ha ? ca : cb;
                    if (ha && hb && half_area(2 * k + 2) > half_area(2 * k + 1)) pick = cb;
                    if (taken[pick]) return MIPT_ERR_BVH;                            // a child pair with two parents: not a tree
                    if (order.size() + 3 > (size_t)cap) return MIPT_ERR_SCENE_LIMIT;
                    if (order.size() & 1u) order.push_back(0xffffffffu);
                    order.push_back(k); order.push_back(pick);
                    taken[pick] = 1;
                }
                if (lone.size() > (size_t)cap) return MIPT_ERR_SCENE_LIMIT;
            }
            for (uint32_t k : level) {
                uint32_t ca = 0, cb = 0;
                const bool ha = child_pair(k, 0, &ca), hb = child_pair(k, 1, &cb);
                if (ha && hb) { couples.push_back(ca); couples.push_back(cb); }
                else if (ha) singles.push_back(ca);
                else if (hb) singles.push_back(cb);
            }
            level = couples;
            level.insert(level.end(), singles.begin(), singles.end());
            depth++;
        }
        if (order.size() & 1u) order.push_back(0xffffffffu);
        if (order.size() + lone.size() > (size_t)cap) return MIPT_ERR_SCENE_LIMIT;
        for (uint32_t k : lone) order.push_back(k);      // level order: sibling pairs that are both free are neighbours (and mostly line mates)
        if (order.size() > cap) return MIPT_ERR_SCENE_LIMIT;
        for (size_t j = 0; j < order.size(); j++) order_out[j] = order[j];
        *n_records_out = (uint32_t)order.size();
        return MIPT_OK;
    } catch (const std::exception &) {
        return MIPT_ERR_INVALID_ARG;
    }
}


// ---- slots of the device's intersection stream: slot_out[i] = index of triangle i's 64-B record (two records per 128-B line) ----
// The memory side moves whole lines, a leaf of the binned-SAH tree holds 1 or 2 triangles almost always (avg 1.32 on the 10 M-triangle
// scene) and the two leaves of a pair are usually tested one after the other.  The stream's order is free -- a leaf only needs its own
// triangles consecutive -- so: first every "double" gets a line to itself (a 2-triangle leaf, or the two 1-triangle leaves of one
// pair, whose triangles are neighbours in the reference order: bvh.rs:99-115 partitions a node's range in place), then all remaining
// triangles follow in the reference order.  The record carries the triangle's reference index, which is what a hit reports.
// Measured with the parent+child pair lines: 10.55 -> 10.04 line fills per ray (profiles/r3_layout_ab.csv).
// The caller has validated the nodes (leaves partition [0, n_tris)).
int mipt::tri_slots(const MiptNode *nodes, uint32_t n_nodes, uint32_t n_tris, uint32_t *slot_out, uint32_t *n_slots_out) {
    if (!nodes || !slot_out || !n_slots_out || (n_nodes & 1u) == 0u) return MIPT_ERR_INVALID_ARG;
    try {
        const uint32_t n_pairs = (n_nodes - 1u) / 2u;
        uint32_t next = 0;
        auto in_range = [&](const MiptNode &n) { return n.num_tris == 0u || (uint64_t)n.first_tri_or_child + n.num_tris <= n_tris; };
        std::vector<uint8_t> placed(n_tris, 0);
        for (uint32_t k = 0; k < n_pairs; k++) {
            const MiptNode &l = nodes[2 * k + 1], &r = nodes[2 * k + 2];
            if (!in_range(l) || !in_range(r)) return MIPT_ERR_BVH;
            if (l.num_tris == 1u && r.num_tris == 1u && r.first_tri_or_child == l.first_tri_or_child + 1u) {
                slot_out[l.first_tri_or_child] = next++; slot_out[r.first_tri_or_child] = next++;
                placed[l.first_tri_or_child] = placed[r.first_tri_or_child] = 1;
                continue;
            }
            for (const MiptNode *n : {&l, &r})
                if (n->num_tris == 2u) {
                    slot_out[n->first_tri_or_child] = next++; slot_out[n->first_tri_or_child + 1u] = next++;
                    placed[n->first_tri_or_child] = placed[n->first_tri_or_child + 1u] = 1;
                }
        }
        for (uint32_t i = 0; i < n_tris; i++)
            if (!placed[i]) slot_out[i] = next++;
        *n_slots_out = next;
        return MIPT_OK;
    } catch (const std::exception &) {
        return MIPT_ERR_INVALID_ARG;
    }
}
