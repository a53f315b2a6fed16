// TEST HARNESS (CPU, g++): the emitter interval table of gfxexp_amd/csrc/emitter_spans.h against the
// oracle's restatement of the reference's three nested DiscreteDistribution1D::sample calls
// (oracle/orc_shared.h, common/common_shared.h:209-247; call sequence restir_di_shared.h:366-415).
//
// The harness builds random and adversarial three-level emitter distributions with the serial-order CDFs
// of lights.hip, runs the product's interval construction (span_record_interval / span_guide_entry, the same
// functions the HIP kernels call), and checks that span_lookup returns exactly the record and the density
// the three searches produce: at every interval end +- 1 ulp, on the PCG32 grid and on arbitrary bit patterns.
// Prints "ok <checks>" or the first mismatch.  Driven by tests/test_emitter_spans.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../gfxexp_amd/csrc/emitter_spans.h"
#include "../../oracle/orc_shared.h"

using namespace gfx;

struct Dist {   // serial-order exclusive scan, integral = CDF[n-1] + w[n-1] (lights.hip, common_host.cpp:171-198)
    std::vector<float> w, cdf, prob;
    float integral = 0.0f;
    void build() {
        cdf.resize(w.size()); prob.resize(w.size());
        float acc = 0.0f, last = 0.0f, lastW = 0.0f;
        for (size_t i = 0; i < w.size(); ++i) { cdf[i] = acc; last = acc; lastW = w[i]; acc += w[i]; }
        integral = w.empty() ? 0.0f : last + lastW;
        for (size_t i = 0; i < w.size(); ++i) prob[i] = w[i] / integral;
    }
    orc::DiscreteDistribution1D view() const {
        orc::DiscreteDistribution1D d;
        d.weights = w.data(); d.CDF = cdf.data(); d.integralValue = integral; d.numValues = static_cast<uint32_t>(w.size());
        return d;
    }
};

struct Group { bool emitter; Dist prims; uint32_t recBase; };
struct Inst { std::vector<Group> groups; Dist geomDist; float scale; };
struct Scene {
    std::vector<Inst> insts;
    Dist instDist;
    uint32_t numRecs = 0;
    std::vector<float> primProb, twoOverLenNg;   // per record
};

// the reference's selection (restir_di_shared.h:366-415): -1 = early out
static int64_t locate_reference(const Scene& sc, float ul, float* density) {
    float lightProb = 1.0f;
    float instProb, uGeomInst;
    const uint32_t i = sc.instDist.view().sample(ul, &instProb, &uGeomInst);
    lightProb *= instProb;
    if (instProb == 0.0f) return -1;
    const Inst& inst = sc.insts[i];
    float geomProb, uPrim;
    const uint32_t k = inst.geomDist.view().sample(uGeomInst, &geomProb, &uPrim);
    lightProb *= geomProb;
    if (geomProb == 0.0f) return -1;
    const Group& g = inst.groups[k];
    float primProb;
    const uint32_t t = g.prims.view().sample(uPrim, &primProb);
    lightProb *= primProb;
    const uint32_t rec = g.recBase + t;
    *density = lightProb * sc.twoOverLenNg[rec];
    return rec;
}

static float random_weight(std::mt19937& gen, int flavour) {
    std::uniform_real_distribution<float> u(0.0f, 1.0f);
    switch (flavour) {
    case 0: return u(gen) + 0.01f;                                   // benign
    case 1: return std::exp(60.0f * (u(gen) - 0.5f));                // ~26 decades
    case 2: return u(gen) < 0.4f ? 0.0f : u(gen);                    // many zeros
    case 3: return u(gen) < 0.5f ? 1e-30f * u(gen) : 1e+3f * u(gen); // absorbed tiny weights
    default: return 1.0f;                                            // ties
    }
}

static Scene make_scene(std::mt19937& gen, int flavour, uint32_t maxInsts) {
    Scene sc;
    const uint32_t ni = 1 + gen() % maxInsts;
    for (uint32_t i = 0; i < ni; ++i) {
        Inst inst;
        inst.scale = 0.25f + 2.0f * (gen() % 1000) / 1000.0f;
        const bool emitterInst = gen() % 4 != 0;
        const uint32_t n2 = 1 + gen() % 4;
        for (uint32_t k = 0; k < n2; ++k) {
            Group g;
            g.emitter = emitterInst && (gen() % 3 != 0);
            g.recBase = 0xFFFFFFFFu;
            if (g.emitter) {
                const uint32_t n3 = 1 + gen() % 24;
                for (uint32_t t = 0; t < n3; ++t) g.prims.w.push_back(random_weight(gen, flavour));
                g.prims.build();
                g.recBase = sc.numRecs;
                sc.numRecs += n3;
                for (uint32_t t = 0; t < n3; ++t) {
                    sc.primProb.push_back(g.prims.w[t] / g.prims.integral);
                    sc.twoOverLenNg.push_back(0.5f + (gen() % 1000) / 100.0f);
                }
            }
            inst.geomDist.w.push_back(g.emitter ? g.prims.integral : 0.0f);   // compute_light_probs.cu:86-93
            inst.groups.push_back(std::move(g));
        }
        inst.geomDist.build();
        bool any = false;
        for (const Group& g : inst.groups) any = any || g.emitter;
        sc.instDist.w.push_back(any ? inst.scale * inst.scale * inst.geomDist.integral : 0.0f);   // :134-142
        sc.insts.push_back(std::move(inst));
    }
    sc.instDist.build();
    return sc;
}

struct Table { std::vector<EmitterSpan> spans; std::vector<SpanGuide> guide; uint32_t cells; bool usable; };

// host mirror of lights.hip: k_span_inst_begin, k_span_records, k_span_finish (without its self-check), k_span_guide
static Table build_table(const Scene& sc) {
    Table tb;
    tb.usable = sc.instDist.integral > 0.0f && sc.instDist.integral < INFINITY && sc.numRecs > 0;
    tb.spans.resize(sc.numRecs);
    const uint32_t ni = static_cast<uint32_t>(sc.insts.size());
    std::vector<uint32_t> instBegin(ni + 1);
    for (uint32_t i = 0; i < ni; ++i) {
        SpanInstPred pred; pred.integral = sc.instDist.integral; pred.i = i; pred.cdfAtI = sc.instDist.cdf[i];
        instBegin[i] = span_bisect(0u, kSpanBitsEnd, pred);
    }
    instBegin[ni] = kSpanBitsEnd;
    for (uint32_t i = 0; i < ni; ++i) {
        const Inst& inst = sc.insts[i];
        SpanRecordKey key;
        key.integral1 = sc.instDist.integral;
        key.lo1 = sc.instDist.cdf[i];
        key.hi1 = i + 1 < ni ? sc.instDist.cdf[i + 1] : key.integral1;
        key.integral2 = inst.geomDist.integral;
        key.n2 = static_cast<uint32_t>(inst.groups.size());
        const float instProb = sc.instDist.prob[i];
        for (uint32_t k = 0; k < key.n2; ++k) {
            const Group& g = inst.groups[k];
            if (g.recBase == 0xFFFFFFFFu) continue;
            key.k = k;
            key.lo2 = inst.geomDist.cdf[k];
            key.hi2 = k + 1 < key.n2 ? inst.geomDist.cdf[k + 1] : key.integral2;
            key.integral3 = g.prims.integral;
            const float geomProb = inst.geomDist.prob[k];
            const bool earlyOut = instProb == 0.0f || geomProb == 0.0f;
            const uint32_t n3 = static_cast<uint32_t>(g.prims.w.size());
            for (uint32_t t = 0; t < n3; ++t) {
                key.t = t; key.cdf3AtT = g.prims.cdf[t];
                uint32_t b, e;
                span_record_interval(key, instBegin[i], instBegin[i + 1], earlyOut, t + 1 == n3, b, e);
                EmitterSpan s;
                s.begin = span_float(b); s.end = span_float(e);
                s.density = ((instProb * geomProb) * sc.primProb[g.recBase + t]) * sc.twoOverLenNg[g.recBase + t];
                s.instSlot = i;
                tb.spans[g.recBase + t] = s;
            }
        }
    }
    for (uint32_t e = 0; e < sc.numRecs; ++e) {
        const uint32_t nextBegin = e + 1 < sc.numRecs ? span_bits(tb.spans[e + 1].begin) : kSpanBitsEnd;
        if (span_bits(tb.spans[e].end) == kSpanPending) tb.spans[e].end = span_float(nextBegin);
        const uint32_t b = span_bits(tb.spans[e].begin), en = span_bits(tb.spans[e].end);
        if (!(b <= en && en <= nextBegin && en <= kSpanBitsEnd)) tb.usable = false;
    }
    uint32_t cells = 256;
    while (cells < GFX_SPAN_CELLS_PER_REC * sc.numRecs && cells < (1u << 22)) cells *= 2;
    tb.cells = cells;
    tb.guide.resize(cells);
    for (uint32_t c = 0; c < cells; ++c) tb.guide[c] = span_guide_entry(tb.spans.data(), sc.numRecs, cells, c);
    return tb;
}

static unsigned long long g_checks = 0;
static bool check(const Scene& sc, const Table& tb, uint32_t ulBits, const char* what) {
    if (ulBits >= kSpanBitsEnd) return true;
    const float ul = span_float(ulBits);
    float refDensity = 0.0f;
    const int64_t ref = locate_reference(sc, ul, &refDensity);
    EmitterSpan s;
    const int32_t got = span_lookup(tb.spans.data(), sc.numRecs, tb.guide.data(), tb.cells, ul, s);
    ++g_checks;
    if (ref != got || (ref >= 0 && span_bits(refDensity) != span_bits(s.density) && !(refDensity != refDensity && s.density != s.density))) {
        std::printf("MISMATCH (%s): ul=%.9g (0x%08x) reference rec=%lld density=%.9g, table rec=%d density=%.9g\n",
                    what, ul, ulBits, static_cast<long long>(ref), refDensity, got, got >= 0 ? s.density : 0.0f);
        return false;
    }
    return true;
}

int main(int argc, char** argv) {
    const uint32_t numScenes = argc > 1 ? static_cast<uint32_t>(std::atoi(argv[1])) : 200;
    const uint32_t numRandom = argc > 2 ? static_cast<uint32_t>(std::atoi(argv[2])) : 20000;
    std::mt19937 gen(20240917u);
    uint32_t usableTables = 0;
    for (uint32_t sIdx = 0; sIdx < numScenes; ++sIdx) {
        const int flavour = static_cast<int>(sIdx % 5);
        const Scene sc = make_scene(gen, flavour, sIdx % 7 == 0 ? 300u : 12u);
        const Table tb = build_table(sc);
        if (!tb.usable) continue;   // the product then runs the three searches themselves
        ++usableTables;
        for (uint32_t e = 0; e < sc.numRecs; ++e) {
            const uint32_t b = span_bits(tb.spans[e].begin), en = span_bits(tb.spans[e].end);
            for (int d = -2; d <= 2; ++d) {
                if (!(d < 0 && b < static_cast<uint32_t>(-d)) && !check(sc, tb, b + d, "begin")) return 1;
                if (!(d < 0 && en < static_cast<uint32_t>(-d)) && !check(sc, tb, en + d, "end")) return 1;
            }
        }
        for (uint32_t r = 0; r < numRandom; ++r) {
            if (!check(sc, tb, span_bits(orc::bits2f((gen() >> 9) | 0x3f800000u) - 1.0f), "pcg grid")) return 1;   // PCG32 float mapping
            if (!check(sc, tb, gen() % kSpanBitsEnd, "bit pattern")) return 1;
        }
        if (!check(sc, tb, 0u, "zero") || !check(sc, tb, 0x3F800000u, "one") || !check(sc, tb, 0x3F7FFFFFu, "below one")) return 1;
    }
    std::printf("ok %llu checks, %u of %u tables usable\n", g_checks, usableTables, numScenes);
    return usableTables * 2 > numScenes ? 0 : 2;
}
