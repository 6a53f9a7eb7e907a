// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// CPU restatement of the reference's superbubble preprocessing:
//   rev_postorder_nodes   /root/reference/src/graphs/tools.rs:5-37
//   SuperbubbleFinder     /root/reference/src/bubbles/finder.rs:15-178
//   BubbleIndex           /root/reference/src/bubbles/index.rs:33-199
// Pinned by the reference's own known-answer tests (finder.rs:188-218,
// index.rs:231-318), transcribed in tests/test_oracle_kat.py.
#pragma once
#include <cstdint>
#include <deque>
#include <limits>
#include <stdexcept>
#include <utility>
#include <vector>

#include "graph.hpp"

namespace poa_oracle {

// tools.rs:5-37
inline std::vector<uint32_t> rev_postorder_nodes(const Graph& g) {
    std::vector<uint32_t> ordered;
    ordered.reserve(g.node_count_with_start_and_end());
    std::vector<std::pair<uint32_t, size_t>> stack;
    stack.push_back({g.start, 0});
    std::vector<uint8_t> visited(g.node_count_with_start_and_end(), 0);
    // NB: the reference never inserts the start node into `visited`; harmless in a DAG.
    while (!stack.empty()) {
        auto& top = stack.back();
        bool pushed = false;
        while (top.second < g.succ[top.first].size()) {
            uint32_t child = g.succ[top.first][top.second++];
            if (!visited[child]) {
                visited[child] = 1;
                stack.push_back({child, 0});
                pushed = true;
                break;
            }
        }
        if (!pushed) {
            ordered.push_back(stack.back().first);
            stack.pop_back();
        }
    }
    std::reverse(ordered.begin(), ordered.end());
    return ordered;
}

struct SuperbubbleFinder {
    const Graph& graph;
    std::vector<size_t> rev_postorder;         // node -> order
    std::vector<uint32_t> inv_rev_postorder;   // order -> node
    std::vector<int64_t> out_parent, out_child;

    // finder.rs:30-66
    explicit SuperbubbleFinder(const Graph& g) : graph(g) {
        inv_rev_postorder = rev_postorder_nodes(g);
        rev_postorder.assign(inv_rev_postorder.size(), 0);
        for (size_t po = 0; po < inv_rev_postorder.size(); ++po)
            rev_postorder.at(inv_rev_postorder[po]) = po;
        size_t n = g.node_count_with_start_and_end();
        out_parent.assign(n, -1);
        out_child.assign(n, std::numeric_limits<int64_t>::max());
        for (uint32_t v = 0; v < n; ++v) {
            int64_t mn = -1; bool any = false;
            for (uint32_t p : g.pred[v]) {
                int64_t r = (int64_t)rev_postorder.at(p);
                if (!any || r < mn) mn = r;
                any = true;
            }
            out_parent[v] = any ? mn : -1;
            int64_t mx = 0; any = false;
            for (uint32_t s : g.succ[v]) {
                int64_t r = (int64_t)rev_postorder.at(s);
                if (!any || r > mx) mx = r;
                any = true;
            }
            out_child[v] = any ? mx : std::numeric_limits<int64_t>::max();
        }
    }

    // finder.rs:115-178, run to exhaustion: (entrance, exit) pairs in yield order.
    std::vector<std::pair<uint32_t, uint32_t>> find_all() const {
        std::vector<std::pair<uint32_t, uint32_t>> out;
        size_t n = graph.node_count_with_start_and_end();
        std::vector<int64_t> opm(n, 0);
        std::vector<uint8_t> has(n, 0);
        auto get = [&](uint32_t k) -> int64_t {
            if (!has[k]) throw std::runtime_error("superbubble finder: out_parent_map key missing (reference would panic)");
            return opm[k];
        };
        auto set = [&](uint32_t k, int64_t v) { opm[k] = v; has[k] = 1; };
        std::vector<uint32_t> stack;
        bool have_cand = false; uint32_t cand = 0;
        auto pop_cand = [&]() {
            if (stack.empty()) { have_cand = false; }
            else { cand = stack.back(); stack.pop_back(); have_cand = true; }
        };
        for (size_t curr = n; curr-- > 0;) {
            bool ret = false; std::pair<uint32_t, uint32_t> to_return{0, 0};
            uint32_t nn = inv_rev_postorder.at(curr);
            int64_t furthest_child = out_child[nn];
            if (furthest_child == (int64_t)curr + 1) {
                if (have_cand) stack.push_back(cand);
                cand = inv_rev_postorder.at(curr + 1); have_cand = true;
            } else {
                while (have_cand) {
                    uint32_t candidate = cand;
                    if (furthest_child <= (int64_t)rev_postorder[candidate]) break;
                    pop_cand();
                    if (have_cand) {
                        int64_t nv = std::min(get(candidate), get(cand));
                        set(cand, nv);
                    }
                }
            }
            if (have_cand) {
                uint32_t candidate = cand;
                if ((size_t)get(candidate) == curr) {
                    to_return = {nn, candidate}; ret = true;
                    pop_cand();
                    if (have_cand) {
                        int64_t nv = std::min(get(candidate), get(cand));
                        set(cand, nv);
                    }
                }
            }
            set(nn, out_parent[nn]);
            if (have_cand) {
                int64_t nv = std::min(get(nn), get(cand));
                set(cand, nv);
            }
            if (ret) out.push_back(to_return);
        }
        return out;
    }
};

struct NodeBubbleMap {  // index.rs:202-207
    uint32_t bubble_exit;
    size_t min_dist_to_exit;
    size_t max_dist_to_exit;
};

struct BubbleIndex {  // index.rs:33-45
    std::vector<uint8_t> is_entrance_v, is_exit_v;
    std::vector<std::vector<NodeBubbleMap>> node_bubble_map;
    std::vector<std::pair<size_t, size_t>> dist_to_end;  // (min, max)

    // index.rs:51-156
    explicit BubbleIndex(const Graph& g) {
        SuperbubbleFinder finder(g);
        size_t n = g.node_count_with_start_and_end();
        is_entrance_v.assign(n, 0); is_exit_v.assign(n, 0);
        for (auto& pr : finder.find_all()) {
            is_entrance_v[pr.first] = 1;
            is_exit_v[pr.second] = 1;
        }
        node_bubble_map.assign(n, {});
        dist_to_end.assign(n, {0, 0});

        using BStack = std::vector<std::pair<size_t, uint32_t>>;
        struct Item { uint32_t node; size_t dist; BStack bstack; };
        std::deque<Item> queue;
        BStack init;
        if (is_exit_v[g.end]) init.push_back({0, g.end});
        queue.push_back({g.end, 0, init});
        std::vector<uint8_t> visited(n, 0);
        visited[g.end] = 1;
        while (!queue.empty()) {
            Item it = std::move(queue.front());
            queue.pop_front();
            for (auto& b : it.bstack)
                node_bubble_map[it.node].push_back({b.second, it.dist - b.first, 0});
            dist_to_end[it.node].first = it.dist;
            for (uint32_t p : g.pred[it.node]) {
                if (visited[p]) continue;
                size_t nd = it.dist + 1;
                BStack nb = it.bstack;
                if (is_entrance_v[p]) {
                    if (nb.empty()) throw std::runtime_error("bubble index: empty bubble stack (reference would panic)");
                    auto top = nb.back(); nb.pop_back();
                    node_bubble_map[p].push_back({top.second, nd - top.first, 0});
                }
                if (is_exit_v[p]) nb.push_back({nd, p});
                visited[p] = 1;
                queue.push_back({p, nd, std::move(nb)});
            }
        }
        for (size_t i = finder.inv_rev_postorder.size(); i-- > 0;) {
            uint32_t v = finder.inv_rev_postorder[i];
            size_t mx = 0;
            for (uint32_t s : g.succ[v]) mx = std::max(mx, dist_to_end[s].second + 1);
            dist_to_end[v].second = mx;
            for (auto& b : node_bubble_map[v])
                b.max_dist_to_exit = mx - dist_to_end[b.bubble_exit].second;
        }
    }

    bool is_exit(uint32_t v) const { return is_exit_v[v]; }
    bool is_entrance(uint32_t v) const { return is_entrance_v[v]; }
    bool node_is_part_of_bubble(uint32_t v) const { return !node_bubble_map[v].empty(); }
    size_t get_min_dist_to_end(uint32_t v) const { return dist_to_end[v].first; }
    size_t get_max_dist_to_end(uint32_t v) const { return dist_to_end[v].second; }
};

}  // namespace poa_oracle
