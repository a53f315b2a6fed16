// bvhlab -- CPU laboratory for the BVH8 builder of gfxexp_amd/csrc/lbvh.hip (diagnostic tooling, not product code).
//
// Rebuilds the product's pipeline on the host (63-bit Morton codes -> radix tree -> bottom-up boxes + the wide-node
// SAH dynamic program -> top-down collapse into 8-wide nodes with octant slots and 8-bit child boxes) and counts, for
// a file of rays, exactly what the traversal kernel would fetch: one node fetch per visited wide node, one triangle
// fetch per hit leaf slot, children in (slot XOR ray octant) order, closest-hit and any-hit semantics of bvh8.hip.h.
// Builder variants are switches, so a change can be judged on the bench scene's rays without a GPU:
//     bvhlab <dir with tris.bin rays_closest.bin rays_any.bin> [key=value ...]
//       split=<beta>      triangle pre-splitting budget (references added / triangles), 0 = off
//       splitmode=karras  spatial-median splits prioritised as in Karras & Aila 2013, sec. 4
//       cprim=<c>         cost of a triangle test relative to a node visit in the collapse DP
//       builder=lbvh|sah  binary tree: Morton radix tree | top-down binned SAH (quality yardstick)
//       order=octant|dist child order: slot XOR octant | true distance order (yardstick for the ordering loss)
#include <omp.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

struct V3 { float x, y, z; };
static inline V3 operator+(V3 a, V3 b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static inline V3 operator*(float s, V3 a) { return { s * a.x, s * a.y, s * a.z }; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
static inline float comp(const V3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }
static inline float& comp(V3& v, int a) { return a == 0 ? v.x : a == 1 ? v.y : v.z; }

struct Box {
    V3 lo{ INFINITY, INFINITY, INFINITY }, hi{ -INFINITY, -INFINITY, -INFINITY };
    void grow(V3 p) { lo = { std::min(lo.x, p.x), std::min(lo.y, p.y), std::min(lo.z, p.z) }; hi = { std::max(hi.x, p.x), std::max(hi.y, p.y), std::max(hi.z, p.z) }; }
    void grow(const Box& b) { grow(b.lo); grow(b.hi); }
    float half_area() const { const V3 d = hi - lo; return d.x * d.y + d.y * d.z + d.z * d.x; }
    bool valid() const { return lo.x <= hi.x && lo.y <= hi.y && lo.z <= hi.z; }
};
static Box isect(const Box& a, const Box& b) {
    Box r;
    r.lo = { std::max(a.lo.x, b.lo.x), std::max(a.lo.y, b.lo.y), std::max(a.lo.z, b.lo.z) };
    r.hi = { std::min(a.hi.x, b.hi.x), std::min(a.hi.y, b.hi.y), std::min(a.hi.z, b.hi.z) };
    return r;
}

struct Tri { V3 a, b, c; };
struct Ref { Box box; uint32_t tri; };

struct Options {
    float split = 0.0f;
    float cprim = 1.0f;
    std::string builder = "lbvh";
    std::string order = "octant";
    std::string slots = "auction";
    int maxleaf = 1;
    int sahbins = 16;
    int verbose = 0;
};

// ------------------------------------------------------------------ pre-splitting (Karras & Aila 2013, sec. 4)
struct SplitCtx { V3 lo, ext; };
// coarsest spatial-median plane (over the three axes) crossing the box: returns level (1 = scene middle), axis, position
static bool important_plane(const SplitCtx& sc, const Box& b, int& level, int& axis, float& pos) {
    level = 1000;
    for (int a = 0; a < 3; ++a) {
        const float e = comp(sc.ext, a);
        if (!(e > 0)) continue;
        const double l = (comp(b.lo, a) - comp(sc.lo, a)) / e, h = (comp(b.hi, a) - comp(sc.lo, a)) / e;
        if (!(h > l)) continue;
        // smallest i such that some j * 2^-i lies strictly inside (l, h)
        for (int i = 1; i <= 30; ++i) {
            const double s = std::ldexp(1.0, i);
            const double j = std::floor(l * s) + 1.0;
            if (j / s < h) {
                if (i < level) { level = i; axis = a; pos = static_cast<float>(comp(sc.lo, a) + e * (j / s)); }
                break;
            }
        }
    }
    return level < 1000;
}
static float ideal_area(const Tri& t) {
    const V3 n = cross(t.b - t.a, t.c - t.a);
    return 0.5f * (std::fabs(n.x) + std::fabs(n.y) + std::fabs(n.z));   // half-area units like Box::half_area
}
static float split_priority(const SplitCtx& sc, const Tri& t, const Box& b) {
    int level, axis; float pos;
    if (!important_plane(sc, b, level, axis, pos)) return 0.0f;
    const float excess = b.half_area() - ideal_area(t);
    if (!(excess > 0)) return 0.0f;
    return std::cbrt(std::ldexp(1.0f, -level) * excess);
}
// boxes of the two parts of triangle t (clipped to `b`) on either side of the axis plane
static void clip_boxes(const Tri& t, const Box& b, int axis, float pos, Box& left, Box& right) {
    const V3 v[3] = { t.a, t.b, t.c };
    left = Box(); right = Box();
    for (int i = 0; i < 3; ++i) {
        const V3 p = v[i], q = v[(i + 1) % 3];
        const float pp = comp(p, axis), qq = comp(q, axis);
        if (pp <= pos) left.grow(p);
        if (pp >= pos) right.grow(p);
        if ((pp < pos && qq > pos) || (pp > pos && qq < pos)) {
            const float s = (pos - pp) / (qq - pp);
            V3 x = p + s * (q - p);
            comp(x, axis) = pos;
            left.grow(x); right.grow(x);
        }
    }
    left = isect(left, b); right = isect(right, b);
    comp(left.hi, axis) = std::min(comp(left.hi, axis), pos);
    comp(right.lo, axis) = std::max(comp(right.lo, axis), pos);
}
static void split_rec(const SplitCtx& sc, const Tri& t, uint32_t ti, const Box& b, int splits, std::vector<Ref>& out) {
    int level, axis; float pos;
    if (splits <= 0 || !important_plane(sc, b, level, axis, pos)) { out.push_back({ b, ti }); return; }
    Box l, r;
    clip_boxes(t, b, axis, pos, l, r);
    if (!l.valid() || !r.valid()) { out.push_back({ b, ti }); return; }
    const float pl = split_priority(sc, t, l), pr = split_priority(sc, t, r);
    const int rest = splits - 1;
    int sl = (pl + pr > 0) ? static_cast<int>(std::floor(rest * pl / (pl + pr) + 0.5f)) : rest / 2;
    sl = std::max(0, std::min(rest, sl));
    split_rec(sc, t, ti, l, sl, out);
    split_rec(sc, t, ti, r, rest - sl, out);
}
static std::vector<Ref> make_refs(const std::vector<Tri>& tris, const Options& opt) {
    std::vector<Ref> refs;
    Box scene;
    std::vector<Box> boxes(tris.size());
    for (size_t i = 0; i < tris.size(); ++i) { Box b; b.grow(tris[i].a); b.grow(tris[i].b); b.grow(tris[i].c); boxes[i] = b; scene.grow(b); }
    if (opt.split <= 0) {
        refs.resize(tris.size());
        for (size_t i = 0; i < tris.size(); ++i) refs[i] = { boxes[i], static_cast<uint32_t>(i) };
        return refs;
    }
    SplitCtx sc{ scene.lo, scene.hi - scene.lo };
    std::vector<float> pri(tris.size());
    double total = 0;
    for (size_t i = 0; i < tris.size(); ++i) { pri[i] = split_priority(sc, tris[i], boxes[i]); total += pri[i]; }
    const double budget = opt.split * tris.size();
    // D such that sum floor(D * p_i) <= budget (bisection as in the paper)
    double dlo = 0, dhi = budget / std::max(total, 1e-30) * 4;
    for (int it = 0; it < 40; ++it) {
        const double d = 0.5 * (dlo + dhi);
        double s = 0;
        for (float p : pri) s += std::floor(d * p);
        if (s <= budget) dlo = d; else dhi = d;
    }
    refs.reserve(static_cast<size_t>(tris.size() * (1.0 + opt.split)) + 16);
    size_t splitTris = 0;
    for (size_t i = 0; i < tris.size(); ++i) {
        const int s = static_cast<int>(std::floor(dlo * pri[i]));
        if (s > 0) ++splitTris;
        split_rec(sc, tris[i], static_cast<uint32_t>(i), boxes[i], s, refs);
    }
    std::printf("presplit: %zu triangles -> %zu references (+%.1f %%), %zu triangles split\n", tris.size(), refs.size(),
                100.0 * (refs.size() - tris.size()) / tris.size(), splitTris);
    return refs;
}

// ------------------------------------------------------------------ binary tree
struct BinTree {
    std::vector<int> left, right;      // per internal node: child >= 0 internal, < 0 leaf ~ref
    std::vector<Box> box;              // per internal node
    std::vector<uint32_t> count;       // refs under the node
    std::vector<uint32_t> order;       // ref order (leaf i = refs[order[i]])
    int root = 0;
};
static uint64_t spread21(uint32_t v) {
    uint64_t x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
static BinTree build_lbvh(const std::vector<Ref>& refs) {
    const int n = static_cast<int>(refs.size());
    Box scene;
    for (const Ref& r : refs) scene.grow(r.box);
    const V3 e = scene.hi - scene.lo;
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    for (int i = 0; i < n; ++i) {
        const V3 c = 0.5f * (refs[i].box.lo + refs[i].box.hi);
        const float sx = e.x > 0 ? (c.x - scene.lo.x) / e.x : 0, sy = e.y > 0 ? (c.y - scene.lo.y) / e.y : 0, sz = e.z > 0 ? (c.z - scene.lo.z) / e.z : 0;
        auto q = [](float s) { const float v = s * 2097152.0f; return v <= 0 ? 0u : std::min(static_cast<uint32_t>(v), 2097151u); };
        keys[i] = { (spread21(q(sx)) << 2) | (spread21(q(sy)) << 1) | spread21(q(sz)), static_cast<uint32_t>(i) };
    }
    std::sort(keys.begin(), keys.end());
    BinTree t;
    t.order.resize(n);
    for (int i = 0; i < n; ++i) t.order[i] = keys[i].second;
    t.left.assign(std::max(n - 1, 1), 0); t.right.assign(std::max(n - 1, 1), 0);
    // top-down equivalent of the Karras radix tree: split [first, last] at the highest differing bit of (code, index)
    auto delta = [&](int i, int j) -> int {
        const uint64_t x = keys[i].first ^ keys[j].first;
        if (x == 0) return 64 + __builtin_clz(static_cast<uint32_t>(i ^ j));
        return __builtin_clzll(x);
    };
    struct Item { int first, last, node; };
    std::vector<Item> stack;
    int next = 1;
    if (n >= 2) stack.push_back({ 0, n - 1, 0 });
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        const int common = delta(it.first, it.last);
        int split = it.first, step = it.last - it.first;
        do {
            step = (step + 1) >> 1;
            const int ns = split + step;
            if (ns < it.last && delta(it.first, ns) > common) split = ns;
        } while (step > 1);
        auto child = [&](int a, int b) -> int {
            if (a == b) return ~a;
            const int id = next++;
            stack.push_back({ a, b, id });
            return id;
        };
        t.left[it.node] = child(it.first, split);
        t.right[it.node] = child(split + 1, it.last);
    }
    return t;
}

// top-down binned SAH over references (yardstick): the same output shape as build_lbvh
static BinTree build_sah(const std::vector<Ref>& refs, int bins) {
    const int n = static_cast<int>(refs.size());
    BinTree t;
    t.order.resize(n);
    std::iota(t.order.begin(), t.order.end(), 0u);
    t.left.assign(std::max(n - 1, 1), 0); t.right.assign(std::max(n - 1, 1), 0);
    struct Item { int first, last, node; };
    std::vector<Item> stack;
    int next = 1;
    if (n >= 2) stack.push_back({ 0, n - 1, 0 });
    std::vector<Box> binBox(bins); std::vector<int> binCnt(bins);
    std::vector<float> rightArea(bins);
    while (!stack.empty()) {
        const Item it = stack.back(); stack.pop_back();
        Box cb;
        for (int i = it.first; i <= it.last; ++i) { const Ref& r = refs[t.order[i]]; cb.grow(0.5f * (r.box.lo + r.box.hi)); }
        int bestAxis = -1, bestBin = -1; float bestCost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            const float lo = comp(cb.lo, a), ext = comp(cb.hi, a) - lo;
            if (!(ext > 0)) continue;
            for (int b = 0; b < bins; ++b) { binBox[b] = Box(); binCnt[b] = 0; }
            for (int i = it.first; i <= it.last; ++i) {
                const Ref& r = refs[t.order[i]];
                const float c = 0.5f * (comp(r.box.lo, a) + comp(r.box.hi, a));
                const int b = std::min(bins - 1, static_cast<int>((c - lo) / ext * bins));
                binBox[b].grow(r.box); ++binCnt[b];
            }
            Box acc;
            for (int b = bins - 1; b >= 1; --b) { acc.grow(binBox[b]); rightArea[b] = binCnt[b] || acc.valid() ? acc.half_area() : 0; }
            Box accL; int cntL = 0, total = it.last - it.first + 1;
            for (int b = 0; b < bins - 1; ++b) {
                if (binCnt[b]) accL.grow(binBox[b]);
                cntL += binCnt[b];
                if (cntL == 0 || cntL == total) continue;
                const float cost = accL.half_area() * cntL + rightArea[b + 1] * (total - cntL);
                if (cost < bestCost) { bestCost = cost; bestAxis = a; bestBin = b; }
            }
        }
        int mid;
        if (bestAxis < 0) mid = (it.first + it.last) / 2;   // coincident centroids
        else {
            const float lo = comp(cb.lo, bestAxis), ext = comp(cb.hi, bestAxis) - lo;
            auto inLeft = [&](uint32_t ri) {
                const Ref& r = refs[ri];
                const float c = 0.5f * (comp(r.box.lo, bestAxis) + comp(r.box.hi, bestAxis));
                return std::min(bins - 1, static_cast<int>((c - lo) / ext * bins)) <= bestBin;
            };
            mid = static_cast<int>(std::partition(t.order.begin() + it.first, t.order.begin() + it.last + 1, inLeft) - t.order.begin()) - 1;
            if (mid < it.first || mid >= it.last) mid = (it.first + it.last) / 2;
        }
        auto child = [&](int a, int b) -> int {
            if (a == b) return ~a;
            const int id = next++;
            stack.push_back({ a, b, id });
            return id;
        };
        t.left[it.node] = child(it.first, mid);
        t.right[it.node] = child(mid + 1, it.last);
    }
    return t;
}

// ------------------------------------------------------------------ bottom-up boxes + DP (lbvh.hip k_fit)
struct Dp { float cost[8]; uint32_t dec; };
static void dp_leaf(float area, float cprim, float c[8]) { for (int i = 1; i <= 7; ++i) c[i] = area * cprim; }
static uint32_t dp_combine(const float l[8], const float r[8], float area, uint32_t numRefs, uint32_t maxLeaf, float cprim, float out[8]) {
    uint32_t dec = 0;
    float dist[9];
    for (int j = 2; j <= 8; ++j) {
        float best = INFINITY; int bk = 1;
        for (int k = 1; k < j; ++k) {
            if (k > 7 || j - k > 7) continue;
            const float v = l[k] + r[j - k];
            if (v < best) { best = v; bk = k; }
        }
        dist[j] = best;
        if (j == 8) dec |= static_cast<uint32_t>(bk);
        else dec |= static_cast<uint32_t>(bk) << (4 * j);
    }
    const float asNode = area * 1.0f + dist[8];
    const float asLeaf = numRefs <= maxLeaf ? area * cprim * static_cast<float>(numRefs) : INFINITY;
    out[1] = std::min(asLeaf, asNode);
    for (int i = 2; i <= 7; ++i) {
        if (out[i - 1] <= dist[i]) { out[i] = out[i - 1]; dec &= ~(0xFu << (4 * i)); }
        else out[i] = dist[i];
    }
    // bit 31 shares nibble 7 with the i = 7 decision (values 1..6): set it after the loop may have cleared that nibble
    if (asLeaf <= asNode) dec |= 0x80000000u;
    return dec;
}

struct Fitted { std::vector<Dp> dp; };
static Fitted fit(BinTree& t, const std::vector<Ref>& refs, const Options& opt) {
    const int n = static_cast<int>(refs.size());
    const int ni = std::max(n - 1, 0);
    t.box.assign(std::max(ni, 1), Box()); t.count.assign(std::max(ni, 1), 0);
    Fitted f; f.dp.resize(std::max(ni, 1));
    // iterative post-order
    std::vector<std::pair<int, int>> stack;   // node, state
    if (ni) stack.push_back({ 0, 0 });
    while (!stack.empty()) {
        auto& top = stack.back();
        const int node = top.first;
        if (top.second == 0) {
            top.second = 1;
            if (t.left[node] >= 0) stack.push_back({ t.left[node], 0 });
            if (t.right[node] >= 0) stack.push_back({ t.right[node], 0 });
            continue;
        }
        stack.pop_back();
        float lc[8], rc[8];
        Box b; uint32_t cnt = 0;
        auto side = [&](int c, float tab[8]) {
            if (c < 0) { const Box& rb = refs[t.order[~c]].box; b.grow(rb); cnt += 1; dp_leaf(rb.half_area(), opt.cprim, tab); }
            else { b.grow(t.box[c]); cnt += t.count[c]; std::memcpy(tab, f.dp[c].cost, sizeof(float) * 8); }
        };
        side(t.left[node], lc); side(t.right[node], rc);
        t.box[node] = b; t.count[node] = cnt;
        f.dp[node].dec = dp_combine(lc, rc, b.half_area(), cnt, static_cast<uint32_t>(opt.maxleaf), opt.cprim, f.dp[node].cost);
    }
    return f;
}

// ------------------------------------------------------------------ collapse (lbvh.hip k_collapse_level)
struct WideNode {
    Box cbox[8];           // decoded quantised child boxes
    int child[8];          // >= 0 wide node, < 0: leaf, ~first into leafRefs
    uint16_t leafCount[8];
    uint8_t valid = 0, imask = 0;
};
struct Wide { std::vector<WideNode> nodes; std::vector<uint32_t> leafTris; int depth = 0; };

static void leaves_under(const BinTree& t, int ref, std::vector<uint32_t>& out) {
    if (ref < 0) { out.push_back(t.order[~ref]); return; }
    std::vector<int> st{ ref };
    while (!st.empty()) {
        const int nd = st.back(); st.pop_back();
        for (int c : { t.left[nd], t.right[nd] }) { if (c < 0) out.push_back(t.order[~c]); else st.push_back(c); }
    }
}

static Wide collapse(const BinTree& t, const Fitted& f, const std::vector<Ref>& refs, const Options& opt) {
    Wide w;
    const int n = static_cast<int>(refs.size());
    if (n == 0) return w;
    struct Work { int bin; int wide; int depth; };
    std::vector<Work> queue;
    w.nodes.emplace_back();
    if (n == 1) {   // single reference: one node, one leaf
        WideNode& nd = w.nodes[0];
        nd.valid = 1; nd.child[0] = ~0; nd.leafCount[0] = 1; nd.cbox[0] = refs[0].box; w.leafTris.push_back(refs[0].tri); w.depth = 1;
        return w;
    }
    queue.push_back({ 0, 0, 1 });
    for (size_t qi = 0; qi < queue.size(); ++qi) {
        const Work wk = queue[qi];
        w.depth = std::max(w.depth, wk.depth);
        struct Kid { int ref; bool leaf; Box box; };
        Kid kids[8]; int nk = 0;
        {   // expand distribute(wk.bin, 8) along the DP decisions
            int stRef[16], stI[16], sp = 0;
            const int k0 = static_cast<int>(f.dp[wk.bin].dec & 0xFu);
            stRef[sp] = t.right[wk.bin]; stI[sp] = 8 - k0; ++sp;
            stRef[sp] = t.left[wk.bin]; stI[sp] = k0; ++sp;
            while (sp > 0) {
                --sp;
                const int ref = stRef[sp]; int i = stI[sp];
                if (ref < 0) { kids[nk++] = { ref, true, refs[t.order[~ref]].box }; continue; }
                const uint32_t d = f.dp[ref].dec;
                int k = 0;
                while (i > 1 && (k = static_cast<int>((d >> (4 * i)) & (i == 7 ? 0x7u : 0xFu))) == 0) --i;
                if (i == 1) { kids[nk++] = { ref, (d >> 31) != 0, t.box[ref] }; continue; }
                stRef[sp] = t.right[ref]; stI[sp] = i - k; ++sp;
                stRef[sp] = t.left[ref]; stI[sp] = k; ++sp;
            }
        }
        Box nb;
        for (int k = 0; k < nk; ++k) nb.grow(kids[k].box);
        // node frame: power-of-two scale per axis, 8-bit grid
        float scale[3];
        for (int a = 0; a < 3; ++a) {
            const float ext = comp(nb.hi, a) - comp(nb.lo, a);
            uint32_t us; const float q = ext / 255.0f; std::memcpy(&us, &q, 4);
            uint32_t e = (us >> 23) + ((us & 0x7FFFFFu) ? 1u : 0u);
            float s; uint32_t sb = e << 23; std::memcpy(&s, &sb, 4);
            while (e < 254u && comp(nb.lo, a) + 255.0f * s < comp(nb.hi, a)) { ++e; sb = e << 23; std::memcpy(&s, &sb, 4); }
            scale[a] = s;
        }
        // octant slots by greedy auction
        int slotOf[8];
        {
            const V3 nc = 0.5f * (nb.lo + nb.hi);
            float cost[8][8];
            for (int k = 0; k < nk; ++k) {
                const V3 cc = 0.5f * (kids[k].box.lo + kids[k].box.hi) - nc;
                for (int s = 0; s < 8; ++s) cost[k][s] = ((s & 1) ? cc.x : -cc.x) + ((s & 2) ? cc.y : -cc.y) + ((s & 4) ? cc.z : -cc.z);
            }
            uint32_t freeSlots = 0xFFu, freeKids = (1u << nk) - 1u;
            for (int it = 0; it < nk; ++it) {
                float best = -INFINITY; int bk = 0, bs = 0;
                for (int k = 0; k < nk; ++k) {
                    if (!((freeKids >> k) & 1u)) continue;
                    for (int s = 0; s < 8; ++s) {
                        if (!((freeSlots >> s) & 1u)) continue;
                        if (cost[k][s] > best) { best = cost[k][s]; bk = k; bs = s; }
                    }
                }
                slotOf[bk] = bs; freeKids &= ~(1u << bk); freeSlots &= ~(1u << bs);
            }
        }
        WideNode nd;
        for (int k = 0; k < nk; ++k) {
            const int s = slotOf[k];
            nd.valid |= 1u << s;
            Box qb;
            for (int a = 0; a < 3; ++a) {
                const float org = comp(nb.lo, a), lo = comp(kids[k].box.lo, a), hi = comp(kids[k].box.hi, a);
                uint32_t l = 0, h = 1;
                if (scale[a] > 0) {
                    l = std::min(static_cast<uint32_t>(std::max((lo - org) / scale[a], 0.0f)), 254u);
                    h = std::min(static_cast<uint32_t>(std::max((hi - org) / scale[a], 0.0f)) + 1u, 255u);
                }
                while (l > 0 && org + static_cast<float>(l) * scale[a] > lo) --l;
                while (h < 255 && org + static_cast<float>(h) * scale[a] < hi) ++h;
                comp(qb.lo, a) = org + static_cast<float>(l) * scale[a];
                comp(qb.hi, a) = org + static_cast<float>(h) * scale[a];
            }
            nd.cbox[s] = qb;
            if (kids[k].leaf) {
                std::vector<uint32_t> under;
                leaves_under(t, kids[k].ref, under);
                nd.child[s] = ~static_cast<int>(w.leafTris.size());
                nd.leafCount[s] = static_cast<uint16_t>(under.size());
                for (uint32_t r : under) w.leafTris.push_back(r);   // reference index (triangle + clipped box)
            }
            else {
                nd.imask |= 1u << s;
                nd.child[s] = static_cast<int>(w.nodes.size());
                w.nodes.emplace_back();
                queue.push_back({ kids[k].ref, nd.child[s], wk.depth + 1 });
            }
        }
        w.nodes[wk.wide] = nd;
    }
    return w;
}

// ------------------------------------------------------------------ traversal model (bvh8.hip.h semantics)
struct Ray { V3 o; float tmin; V3 d; float tmax; };
struct Stats { unsigned long long nodes = 0, tris = 0, rays = 0, hits = 0; };

struct Tracer {
    const Wide& w; const std::vector<Ref>& refs; const std::vector<Tri>& tris; bool any; bool distOrder;
    V3 o, d, inv; float tmin, tbest; uint32_t oct; bool done; Stats st;
    bool tri_test(const Tri& t, float& tt) const {
        const V3 eAB = t.b - t.a, eCA = t.a - t.c, n = cross(eCA, eAB);
        const V3 e = (1.0f / dot(n, d)) * (t.a - o);
        const V3 i = cross(d, e);
        const float b = dot(i, eCA), c = dot(i, eAB);
        tt = dot(n, e);
        return tt > tmin && b >= 0 && c >= 0 && b + c <= 1;
    }
    void visit(int ni) {
        ++st.nodes;
        const WideNode& nd = w.nodes[ni];
        uint32_t hit = 0; float tn8[8];
        for (int s = 0; s < 8; ++s) {
            if (!((nd.valid >> s) & 1u)) continue;
            const Box& b = nd.cbox[s];
            const float tx0 = (b.lo.x - o.x) * inv.x, tx1 = (b.hi.x - o.x) * inv.x;
            const float ty0 = (b.lo.y - o.y) * inv.y, ty1 = (b.hi.y - o.y) * inv.y;
            const float tz0 = (b.lo.z - o.z) * inv.z, tz1 = (b.hi.z - o.z) * inv.z;
            const float tn = std::max(std::max(std::min(tx0, tx1), std::min(ty0, ty1)), std::max(std::min(tz0, tz1), tmin));
            const float tf = std::min(std::min(std::max(tx0, tx1), std::max(ty0, ty1)), std::min(std::max(tz0, tz1), tbest));
            if (tn <= tf) { hit |= 1u << s; tn8[s] = tn; }
        }
        // leaf children first, ascending slot order
        for (int s = 0; s < 8 && !done; ++s) {
            if (!((hit >> s) & 1u) || ((nd.imask >> s) & 1u)) continue;
            const int first = ~nd.child[s];
            for (int k = 0; k < nd.leafCount[s] && !done; ++k) {
                ++st.tris;
                float tt;
                if (tri_test(tris[refs[w.leafTris[first + k]].tri], tt) && tt < tbest) { tbest = tt; if (any) done = true; }
            }
        }
        if (done) return;
        uint32_t ih = hit & nd.imask;
        if (!distOrder) {
            for (int p = 0; p < 8 && !done; ++p) {
                const int s = p ^ static_cast<int>(oct);
                if ((ih >> s) & 1u) visit(nd.child[s]);
            }
        }
        else {
            while (ih && !done) {
                int bs = -1; float bt = INFINITY;
                for (int s = 0; s < 8; ++s) if (((ih >> s) & 1u) && tn8[s] < bt) { bt = tn8[s]; bs = s; }
                ih &= ~(1u << bs);
                visit(nd.child[bs]);
            }
        }
    }
    void trace(const Ray& r) {
        o = r.o; d = r.d; tmin = r.tmin; tbest = r.tmax; done = false;
        auto safe = [](float v) { return std::fabs(v) < 1e-20f ? std::copysign(1e-20f, v) : v; };
        inv = { 1.0f / safe(d.x), 1.0f / safe(d.y), 1.0f / safe(d.z) };
        oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        ++st.rays;
        if (!(r.tmax > r.tmin) || w.nodes.empty()) return;
        visit(0);
        if (tbest < r.tmax) ++st.hits;
    }
};

static Stats run(const Wide& w, const std::vector<Ref>& refs, const std::vector<Tri>& tris, const std::vector<Ray>& rays, bool any, bool distOrder) {
    Stats total;
#pragma omp parallel
    {
        Tracer tr{ w, refs, tris, any, distOrder };
#pragma omp for schedule(dynamic, 512)
        for (long i = 0; i < static_cast<long>(rays.size()); ++i) tr.trace(rays[i]);
#pragma omp critical
        { total.nodes += tr.st.nodes; total.tris += tr.st.tris; total.rays += tr.st.rays; total.hits += tr.st.hits; }
    }
    return total;
}

template <typename T>
static std::vector<T> read_file(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(1); }
    std::fseek(f, 0, SEEK_END);
    const long bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<T> v(bytes / sizeof(T));
    if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) { std::fprintf(stderr, "short read %s\n", path.c_str()); std::exit(1); }
    std::fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: bvhlab <dir> [key=value ...]\n"); return 1; }
    const std::string dir = argv[1];
    Options opt;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        const size_t eq = a.find('=');
        if (eq == std::string::npos) continue;
        const std::string k = a.substr(0, eq), v = a.substr(eq + 1);
        if (k == "split") opt.split = std::atof(v.c_str());
        else if (k == "cprim") opt.cprim = std::atof(v.c_str());
        else if (k == "builder") opt.builder = v;
        else if (k == "order") opt.order = v;
        else if (k == "maxleaf") opt.maxleaf = std::atoi(v.c_str());
        else if (k == "bins") opt.sahbins = std::atoi(v.c_str());
        else if (k == "verbose") opt.verbose = std::atoi(v.c_str());
    }
    const std::vector<Tri> tris = read_file<Tri>(dir + "/tris.bin");
    const std::vector<Ray> closest = read_file<Ray>(dir + "/rays_closest.bin");
    const std::vector<Ray> anyRays = read_file<Ray>(dir + "/rays_any.bin");
    double t0 = omp_get_wtime();
    const std::vector<Ref> refs = make_refs(tris, opt);
    BinTree bt = opt.builder == "sah" ? build_sah(refs, opt.sahbins) : build_lbvh(refs);
    const Fitted f = fit(bt, refs, opt);
    const Wide w = collapse(bt, f, refs, opt);
    const double buildS = omp_get_wtime() - t0;
    double sahCost = 0;
    {   // SAH cost of the wide tree (node visits + triangle tests, area weighted, relative to the root)
        Box root; for (int s = 0; s < 8; ++s) if ((w.nodes[0].valid >> s) & 1u) root.grow(w.nodes[0].cbox[s]);
        for (const WideNode& nd : w.nodes) for (int s = 0; s < 8; ++s) if ((nd.valid >> s) & 1u) sahCost += nd.cbox[s].half_area() / root.half_area();
    }
    if (opt.verbose) {
        unsigned long long kids = 0, leafSlots = 0, multi = 0, hist[9] = { 0 };
        for (const WideNode& nd : w.nodes) {
            const int nk = __builtin_popcount(nd.valid);
            ++hist[nk]; kids += nk;
            for (int s = 0; s < 8; ++s) if (((nd.valid >> s) & 1u) && !((nd.imask >> s) & 1u)) { ++leafSlots; if (nd.leafCount[s] > 1) ++multi; }
        }
        std::printf("nodes %zu, children/node %.2f, leaf slots %llu (multi-triangle %llu), leaf refs %zu; children histogram:", w.nodes.size(),
                    static_cast<double>(kids) / w.nodes.size(), leafSlots, multi, w.leafTris.size());
        for (int k = 0; k <= 8; ++k) std::printf(" %llu", hist[k]);
        std::printf("\n");
    }
    const bool distOrder = opt.order == "dist";
    const Stats c = run(w, refs, tris, closest, false, distOrder);
    const Stats a = run(w, refs, tris, anyRays, true, distOrder);
    // bytes the three launches of a frame would fetch: 1 closest launch + 2 any-hit launches of these ray sets
    const double frameBytes = (c.nodes + 2.0 * a.nodes) * 80 + (c.tris + 2.0 * a.tris) * 64;
    std::printf("{\"builder\": \"%s\", \"split\": %.2f, \"cprim\": %.2f, \"maxleaf\": %d, \"order\": \"%s\", \"refs\": %zu, \"wide_nodes\": %zu, \"depth\": %d, "
                "\"sah_cost\": %.1f, \"build_s\": %.2f, "
                "\"closest\": {\"nodes_per_ray\": %.3f, \"tris_per_ray\": %.3f, \"hit\": %.3f}, "
                "\"any\": {\"nodes_per_ray\": %.3f, \"tris_per_ray\": %.3f, \"occluded\": %.3f}, \"frame_fetch_bytes_rel\": %.4e}\n",
                opt.builder.c_str(), opt.split, opt.cprim, opt.maxleaf, opt.order.c_str(), refs.size(), w.nodes.size(), w.depth, sahCost, buildS,
                static_cast<double>(c.nodes) / c.rays, static_cast<double>(c.tris) / c.rays, static_cast<double>(c.hits) / c.rays,
                static_cast<double>(a.nodes) / a.rays, static_cast<double>(a.tris) / a.rays, static_cast<double>(a.hits) / a.rays, frameBytes);
    return 0;
}
