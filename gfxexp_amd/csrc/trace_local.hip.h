// trace_local.hip.h -- the traversal of trace.hip as a device function: a wave traces the (at most) 64 rays its own lanes hold.
//
// k_trace is a launch of its own: persistent waves, a ticket queue, idle lanes refilled while the others keep traversing -- what a
// full-HD pass needs to keep 0.83 of the lanes busy.  A SMALL pass (one rank's row band of a multi-GPU frame: ~260 k rays on the
// 262 k lanes the persistent grid has) has nothing to refill with: every lane gets one ray and the launch lasts as long as its
// longest ray.  Three such launches per frame (primary rays, the visibility ray of the selected candidate, the final shadow ray),
// each between two short per-pixel kernels, make a band's frame a chain of eleven launches that each wait for their slowest wave
// (profiles/r04_experiments.txt 3, 8, 13).  With this function the producer of a ray traces it and consumes the result in the same
// kernel: no queue, no launch boundary -- a wave that is done with its rays goes on while its neighbours are still traversing
// (restir.hip k_gbuffer_fused / k_initial_fused / k_shading_fused).  Same arithmetic as k_trace (bvh8.hip.h Traversal), same result:
// closest hits are order independent by the tie rule, any-hit results are one bit.
#pragma once
#include "bvh8.hip.h"
#include "coop_fetch.hip.h"

namespace gfx {

// Entries of HBM stack spill a lane needs behind its kLdsStackDepth LDS entries for a tree `maxDepth` levels deep (host side; lbvh.hip
// refuses trees deeper than kTraceLdsStackDepth + kSpillStackDepth - 1): one entry per level below the LDS part, + 1, at least 4.
inline uint32_t local_spill_depth(uint32_t maxDepth) {
    const uint32_t need = maxDepth + 1 > static_cast<uint32_t>(kLdsStackDepth) ? maxDepth + 1 - static_cast<uint32_t>(kLdsStackDepth) : 0u;
    return need + 1 < 4u ? 4u : need + 1;
}

// EVERY lane of the wave must call (the item fetch is cooperative); a lane without a ray passes want = false.
//   stackLds / stackStride  this lane's LDS column (kLdsStackDepth entries, `stackStride` uint2 apart)
//   stackSpill / spillCap   its HBM spill area of spillCap entries: local_spill_depth() of the tree -- a stack is never deeper than the
//                           tree, so the area is sized by the tree (a few entries), not by the worst case k_trace's fixed grid can afford
//   waveBuf                 256 x 16 B of LDS private to the wave (the cooperative fetch)
//   hint                    closest hit only: a triangle record to test right after the root (k_trace's temporal hint), or an
//                           index >= numTris for none
//   waveSteps               optional out: how many steps the wave took (its longest ray)
template <bool ANY_HIT>
GFX_DEV RayHit trace_wave_local(const DevAccel& accel, bool want, f3 org, f3 dir, float tmin, float tmax, uint2* stackLds, int stackStride,
                                uint2* stackSpill, int spillCap, uint4* waveBuf, int lane, uint32_t hint = 0xFFFFFFFFu, uint32_t* waveSteps = nullptr) {
    LaneStack stack;
    stack.lds = stackLds; stack.ldsStride = stackStride; stack.spill = stackSpill; stack.sp = 0; stack.spillCap = spillCap; stack.ldsDepth = kLdsStackDepth;
    const bool hasNodes = accel.numNodes != 0;
    Traversal tr;
    tr.begin(org, dir, tmin, tmax, stack, hasNodes, scene_max_abs(accel));
    tr.active = want && hasNodes && tmax > tmin;        // an empty interval or an empty scene: a miss (hit.t = tmax, no triangle)
    TraceCounters cnt = { 0, 0, 0 };
    bool first = true;
    uint32_t steps = 0;                                  // wave-uniform: iterations until the wave's last ray has ended
    while (__ballot(tr.active) != 0ull) {
        ++steps;
        uint32_t code = kItemNone;
        if (tr.active) code = tr.next_item(stack, accel.triItemOffset);      // the first item of a ray is the root (begin's one-child group)
        uint4 link;                    // read only by process_node, i.e. under the condition of its load (no default: v_mov per iteration)
        if (code != kItemNone && !(code & kItemTri)) link = reinterpret_cast<const uint4*>(accel.links)[code];   // in flight with the item fetch
        uint4 q0, q1, q2, q3;
        fetch_items(code, accel, waveBuf, lane, q0, q1, q2, q3);
        if (code != kItemNone) {
            if (code & kItemTri) (void)tr.template process_triangle<ANY_HIT, false>((code & 0x7FFFFFFFu) - accel.triItemOffset, q0, q1, q2, q3, accel.tris, cnt);
            else tr.template process_node<false>(q0, q1, q2, q3, link, stack, cnt);
        }
        if (!ANY_HIT && first && tr.active && hint < accel.numTris && tr.triMask == 0u) { tr.triBase = hint; tr.triMask = 0x0101u; }
        first = false;
    }
    if (waveSteps) *waveSteps = steps;
    return tr.hit;
}

} // namespace gfx
