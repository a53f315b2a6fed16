// emitter_spans.h -- the three-level light selection of sampleLight, flattened into one interval table.
//
// The reference picks an emitter triangle from one uniform number ul by three nested
// DiscreteDistribution1D::sample calls (instance -> geometry instance -> primitive,
// restir_di/restir_di_shared.h:366-415; the search itself common/common_shared.h:209-247), each
// remapping ul for the next level.  Every level is a monotone step function of its input and every remap
// (u - lo) / (hi - lo) is monotone in u under IEEE fp32, so the composite map
//     ul  ->  (instance, geometry instance, primitive)
// is a monotone step function of ul in the lexicographic order of its result -- which is the order the
// emitter records (EmitterRec, device_types.h) are stored in.  Hence the set of ul that select record e is
// one interval [begin_e, end_e) of fp32 values, the intervals are disjoint and ascending in e, and the
// interval ends can be found EXACTLY by bisection over the bit patterns of ul with the reference's own
// arithmetic as the predicate.  The light-distribution build (lights.hip) does that once per change of the
// distributions; sample_light (shading.hip.h) then replaces three dependent searches, two divisions and
// ~10 dependent memory round trips per candidate by one guided search in this table.  ul values that fall
// into no interval are exactly those for which the reference returns early with a zero density
// (an instance or geometry instance of probability zero).
//
// Everything here is plain C++ that compiles for the device (hipcc) and for the host (g++, the unit test
// tests/test_emitter_spans.py drives it against the oracle's restatement of the reference search).
#pragma once
#include <stdint.h>

#if defined(__HIP__)   // clang in HIP language mode (.hip sources): usable from kernels and from host code
#define GFX_SPAN_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define GFX_SPAN_HD inline
#endif

namespace gfx {

// One emitter record's interval of ul plus what sample_light needs besides the record itself.
struct EmitterSpan {
    float begin, end;      // ul in [begin, end) selects this record; begin == end: never selected
    float density;         // ((instProb * geomInstProb) * primProb) * (2 / |ng|): the area density of a sample on it
    uint32_t instSlot;
};
static_assert(sizeof(EmitterSpan) == 16, "EmitterSpan must be 16 bytes");

GFX_SPAN_HD uint32_t span_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
GFX_SPAN_HD float span_float(uint32_t u) { return __builtin_bit_cast(float, u); }

// ul ranges over [0, 1] (a remapped ul can round up to 1); "no such ul" is the next float above 1.
constexpr uint32_t kSpanBitsEnd = 0x3F800001u;

// Smallest fp32 in [lo, hi) (as bit patterns of non-negative floats) for which pred holds, hi if none.
// pred must be monotone (false ... false true ... true) over the range.
template <typename Pred>
GFX_SPAN_HD uint32_t span_bisect(uint32_t lo, uint32_t hi, Pred pred) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pred(span_float(mid))) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// "the search returns an index >= i" for DiscreteDistribution1D::sample's loop: the loop finds the largest
// idx with CDF[idx] <= u (CDF monotone, CDF[0] = 0 <= u); with u = NaN every comparison fails and idx = 0.
GFX_SPAN_HD bool span_index_ge(uint32_t i, float cdfAtI, float u) { return i == 0 || cdfAtI <= u; }

// Level 1: the instance search reaches index >= i.
struct SpanInstPred {
    float integral; uint32_t i; float cdfAtI;
    GFX_SPAN_HD bool operator()(float ul) const { return span_index_ge(i, cdfAtI, ul * integral); }
};

// Levels 2 and 3 for a ul that selects instance i at level 1.  All operands are the values the reference
// reads for exactly this (instance, geometry instance, primitive); nothing is loaded inside the bisection.
struct SpanRecordKey {
    // level 1
    float integral1, lo1, hi1;          // integral, CDF[i], CDF[i + 1] (integral for the last entry)
    // level 2 (geometry instance k of n2 inside instance i)
    float integral2, lo2, hi2;          // integral, CDF[k], CDF[k + 1] (integral for the last entry)
    uint32_t k, n2;
    // level 3 (primitive t)
    float integral3, cdf3AtT;
    uint32_t t;

    GFX_SPAN_HD float level2_u(float ul) const {
        const float u1 = ul * integral1;
        const float r1 = (u1 - lo1) / (hi1 - lo1);
        return r1 * integral2;
    }
    // the level-2 search lands past k
    GFX_SPAN_HD bool past_group(float ul) const {
        const float u2 = level2_u(ul);
        return k + 1 < n2 && hi2 <= u2;
    }
    // (k', t') >= (k, t) lexicographically
    GFX_SPAN_HD bool at_or_past_record(float ul) const {
        const float u2 = level2_u(ul);
        if (k + 1 < n2 && hi2 <= u2) return true;
        if (!span_index_ge(k, lo2, u2)) return false;
        const float r2 = (u2 - lo2) / (hi2 - lo2);
        const float u3 = r2 * integral3;
        return span_index_ge(t, cdf3AtT, u3);
    }
};

// "end = the next record's begin": resolved once every begin is known (k_span_finish)
constexpr uint32_t kSpanPending = 0xFFFFFFFFu;

// Interval of one record inside its instance's range [rangeLo, rangeHi) of ul bit patterns.
// earlyOut: the reference returns before reaching a record (instance or geometry-instance probability zero);
// lastOfGroup: last primitive of its geometry instance (its end is where the level-2 search moves on).
GFX_SPAN_HD void span_record_interval(const SpanRecordKey& key, uint32_t rangeLo, uint32_t rangeHi, bool earlyOut, bool lastOfGroup,
                                      uint32_t& beginBits, uint32_t& endBits) {
    const SpanRecordKey kk = key;
    beginBits = span_bisect(rangeLo, rangeHi, [kk](float ul) { return kk.at_or_past_record(ul); });
    endBits = kSpanPending;
    if (earlyOut) endBits = beginBits;
    else if (lastOfGroup) endBits = span_bisect(beginBits, rangeHi, [kk](float ul) { return kk.past_group(ul); });
}

// Guide table entry of one cell.
//   boundary cell (a record's interval begins inside it): a = hi, b = lo -- the answer for any ul of the cell lies in [lo, hi];
//   interior cell (every ul of the cell selects the same record): a = kGuideInterior | record, b = bit pattern of the record's
//   density -- the lookup is done after this one load (no probe, no span load).  With GFX_SPAN_CELLS_PER_REC cells per record at
//   most 1 / GFX_SPAN_CELLS_PER_REC of the ul range lies in boundary cells.
struct alignas(8) SpanGuide { uint32_t a, b; };
constexpr uint32_t kGuideInterior = 0x80000000u;
#ifndef GFX_SPAN_CELLS_PER_REC
#define GFX_SPAN_CELLS_PER_REC 4u
#endif

// Guide cell of a ul (cells is a power of two, so ul * cells is exact).
GFX_SPAN_HD uint32_t span_cell(float ul, uint32_t cells) {
    const uint32_t c = static_cast<uint32_t>(ul * static_cast<float>(cells));
    return c < cells - 1u ? c : cells - 1u;
}

// Smallest ul (bit pattern) that span_cell maps to a cell >= c; kSpanBitsEnd when there is none (c >= cells).
GFX_SPAN_HD uint32_t span_cell_first_ul(uint32_t cells, uint32_t c) {
    if (c >= cells) return kSpanBitsEnd;
    return span_bisect(0u, kSpanBitsEnd, [cells, c](float ul) { return span_cell(ul, cells) >= c; });
}

// Guide entry of cell c: hi = the largest j with cell(begin_j) <= c (begin ascends, span_cell is monotone),
// lo = the same for c - 1; 0 when there is none.  lo == hi means that no interval begins inside the cell; if that record's
// interval also covers every ul of the cell -- begin <= the cell's smallest ul, end >= the next cell's smallest ul (the first
// float above 1 for the last cell) -- the cell is interior.
GFX_SPAN_HD SpanGuide span_guide_entry(const EmitterSpan* __restrict__ spans, uint32_t numSpans, uint32_t cells, uint32_t c) {
    auto last_at_or_before = [&](uint32_t cell) {
        uint32_t lo = 0, hi = numSpans;   // first j in [0, numSpans] with cell(begin_j) > cell
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if (span_cell(spans[mid].begin, cells) <= cell) lo = mid + 1;
            else hi = mid;
        }
        return lo ? lo - 1u : 0u;
    };
    SpanGuide g;
    g.a = last_at_or_before(c);
    g.b = c ? last_at_or_before(c - 1u) : 0u;
    if (g.a == g.b && numSpans != 0 && g.a < kGuideInterior) {
        const EmitterSpan s = spans[g.a];
        const uint32_t first = span_cell_first_ul(cells, c), next = span_cell_first_ul(cells, c + 1u);
        if (span_bits(s.begin) <= first && span_bits(s.end) >= next && first < next) {
            g.b = span_bits(s.density);
            g.a |= kGuideInterior;
        }
    }
    return g;
}

// Index of the record whose interval holds ul, or -1; out.density = its density.  The other members of `out` are set only when
// the answer came out of a boundary cell.
struct alignas(16) SpanWords { uint32_t w[4]; };   // one aligned 16-byte load per span
GFX_SPAN_HD int32_t span_lookup(const EmitterSpan* __restrict__ spans, uint32_t numSpans,
                                const SpanGuide* __restrict__ guide, uint32_t cells, float ul, EmitterSpan& out) {
    if (numSpans == 0) return -1;
    const SpanGuide g = guide[span_cell(ul, cells)];
    if (g.a & kGuideInterior) {
        out.begin = 0.0f; out.end = 0.0f; out.density = span_float(g.b); out.instSlot = 0xFFFFFFFFu;
        return static_cast<int32_t>(g.a & ~kGuideInterior);
    }
    int32_t lo = static_cast<int32_t>(g.b), hi = static_cast<int32_t>(g.a);
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (spans[mid].begin <= ul) lo = mid;
        else hi = mid - 1;
    }
    const SpanWords raw = *reinterpret_cast<const SpanWords*>(spans + lo);
    out = __builtin_bit_cast(EmitterSpan, raw);
    return (ul >= out.begin && ul < out.end) ? lo : -1;
}

} // namespace gfx
