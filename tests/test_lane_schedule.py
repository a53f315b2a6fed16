"""CPU: the lanes of a band renderer's frame program order every conflicting access (no GPU).

gfxh_restir_frame_program says on which lane (stream) each pass and each exchange of a frame is issued and where the frame's stream
waits for another lane; gfxh_restir_render_frame turns that into streams and events (csrc/host/restir_driver.cpp).  This test
restates the happens-before relation those events create -- stream order inside a lane, plus one edge per event -- over three
consecutive frames of an interior band, gives every step its read and write sets (buffer, ping-pong half, row range) from the
definition of the passes, and checks that every pair of steps that touch the same rows of the same buffer with at least one write
is ordered.  A program that moved an exchange to another lane without the wait that goes with it fails here by step name."""
import itertools

import pytest

from gfxexp_amd import api

W, H, WORLD, RADIUS = 1920, 1080, 8, 20


def _overlap(a, b):
    return a[0] < b[1] and b[0] < a[1]


def _accesses(st, frame, band, motion):
    """[(buffer name, (row begin, row end), 'r' | 'w')] of one step of frame `frame`."""
    b, e = band
    h, ph = frame % 2, (frame + 1) % 2
    out = []

    def rows_of(st):
        if st.gapEnd > st.gapBegin:
            return [(st.rowBegin, st.gapBegin), (st.gapEnd, st.rowEnd)]
        return [(st.rowBegin, st.rowEnd)]
    if st.op == api.STEP_RESTIR_PASS:
        for rb, re in rows_of(st):
            if re <= rb:
                continue
            halo = (max(0, rb - RADIUS), min(H, re + RADIUS))
            cur = st.currentReservoirIndex
            if st.pass_ == api.PASS_SETUP_GBUFFERS:
                out += [("gbuffer%d" % h, (rb, re), "w")]
            elif st.pass_ in (api.PASS_INITIAL_RIS, api.PASS_INITIAL_TEMPORAL_BIASED, api.PASS_INITIAL_TEMPORAL_UNBIASED):
                out += [("gbuffer%d" % h, (rb, re), "r"), ("reservoir%d" % cur, (rb, re), "w"), ("rng", (rb, re), "w")]
                if st.pass_ != api.PASS_INITIAL_RIS:
                    m = (max(0, rb - motion), min(H, re + motion))
                    out += [("gbuffer%d" % ph, m, "r"), ("reservoir%d" % (1 - cur), m, "r")]
            elif st.pass_ in (api.PASS_SPATIAL_BIASED, api.PASS_SPATIAL_UNBIASED, api.PASS_SPATIAL_BIASED_AND_SHADING):
                out += [("gbuffer%d" % h, halo, "r"), ("reservoir%d" % cur, halo, "r"), ("reservoir%d" % (1 - cur), (rb, re), "w"), ("rng", (rb, re), "w")]
                if st.pass_ == api.PASS_SPATIAL_BIASED_AND_SHADING:
                    out += [("beauty", (rb, re), "w")]
            elif st.pass_ == api.PASS_SHADING:
                out += [("gbuffer%d" % h, (rb, re), "r"), ("reservoir%d" % st.currentReservoirIndex, (rb, re), "r"), ("beauty", (rb, re), "w"), ("rng", (rb, re), "w")]
            else:
                raise AssertionError("pass %d is not modelled" % st.pass_)
    elif st.op == api.STEP_EXCHANGE_STRIPS:
        n = st.exchangeRows
        send = [(b, min(e, b + n)), (max(b, e - n), e)]
        recv = [(max(0, b - n), b), (e, min(H, e + n))]
        names = []
        if st.buffers & api.BUF_GBUFFERS:
            names.append("gbuffer%d" % h)
        if st.buffers & api.BUF_RESERVOIRS:
            names.append("reservoir%d" % st.reservoirIndex)
        if st.buffers & api.BUF_RNG:
            names.append("rng")
        for name in names:
            out += [(name, r, "r") for r in send] + [(name, r, "w") for r in recv]
    elif st.op == api.STEP_GATHER_BANDS:
        out += [("beauty", (b, e), "r"), ("beauty", (0, b), "w"), ("beauty", (e, H), "w")]
    return [a for a in out if a[1][1] > a[1][0]]


def _schedule(cfg, frames, motion, strip_mode):
    """Nodes (frame, index, step) in issue order and the happens-before edges of the driver's events."""
    unbiased = cfg.renderer == api.RENDERER_UNBIASED
    nodes, edges = [], set()
    last_on = {}                       # lane -> node id of the last operation issued on it (stream order)
    last_res, last_base = 1, 0
    prev_read = None                   # MAIN node behind which the previous G-buffer half is no longer read (evPrevRead)
    gather_node = None                 # lane GATHER node of the previous frame (evGather)
    for frame in range(frames):
        steps, last_res, last_base = api.frame_program(cfg, strip_mode, motion, frame == 0, last_res, last_base, unbiased)
        gb_pass = gb_strips = seam_strips = None
        gb_waited = False
        for k, st in enumerate(steps):
            if st.op in (api.STEP_RESTIR_PASS, api.STEP_EXCHANGE_STRIPS, api.STEP_GATHER_BANDS):
                nid = len(nodes)
                nodes.append((frame, k, st))
                lane = st.lane
                if lane in last_on:
                    edges.add((last_on[lane], nid))                     # stream order
                if lane == api.LANE_GBUFFER and st.op == api.STEP_RESTIR_PASS:
                    if prev_read is not None:
                        edges.add((prev_read, nid))                     # hipStreamWaitEvent(gbStream, evPrevRead)
                    gb_pass = nid
                if lane == api.LANE_GBUFFER and st.op == api.STEP_EXCHANGE_STRIPS:
                    gb_strips = nid
                if lane == api.LANE_MAIN and not gb_waited and gb_pass is not None:
                    edges.add((gb_pass, nid))                           # hipStreamWaitEvent(main, evGbuffer): the kernel, not its strips
                    gb_waited = True
                if lane == api.LANE_SEAM:
                    edges.add((last_on[api.LANE_MAIN], nid))            # evSeamRows: behind the seam rows just queued on MAIN
                    seam_strips = nid
                if lane == api.LANE_GATHER:
                    edges.add((last_on[api.LANE_MAIN], nid))            # evBandDone
                    gather_node = nid
                last_on[lane] = nid
            elif st.op == api.STEP_PREV_GBUFFER_RELEASED:
                prev_read = last_on[api.LANE_MAIN]
            elif st.op in (api.STEP_WAIT_GBUFFER_STRIPS, api.STEP_WAIT_SEAM_STRIPS, api.STEP_WAIT_PREVIOUS_GATHER):
                src = {api.STEP_WAIT_GBUFFER_STRIPS: gb_strips, api.STEP_WAIT_SEAM_STRIPS: seam_strips, api.STEP_WAIT_PREVIOUS_GATHER: gather_node}[st.op]
                if src is not None:
                    # the wait is an operation of its own on MAIN: everything issued on MAIN afterwards is behind `src`
                    nid = len(nodes)
                    nodes.append((frame, k, st))
                    edges.add((src, nid))
                    if api.LANE_MAIN in last_on:
                        edges.add((last_on[api.LANE_MAIN], nid))
                    last_on[api.LANE_MAIN] = nid
    return nodes, edges


def _reachability(n, edges):
    succ = [[] for _ in range(n)]
    for a, b in edges:
        succ[a].append(b)
    reach = [0] * n                      # bit sets; edges only point forward in issue order
    for a in range(n - 1, -1, -1):
        r = 0
        for b in succ[a]:
            r |= (1 << b) | reach[b]
        reach[a] = r
    return reach


def _check(cfg, motion, strip_mode):
    band = api.band_rows(H, WORLD, WORLD // 2)
    cfg.rowBegin, cfg.rowEnd = band
    nodes, edges = _schedule(cfg, 4, motion, strip_mode)
    reach = _reachability(len(nodes), edges)
    acc = [_accesses(st, frame, band, motion) for frame, k, st in nodes]
    problems = []
    for i, j in itertools.combinations(range(len(nodes)), 2):
        if (reach[i] >> j) & 1:
            continue
        for (na, ra, ma), (nb, rb, mb) in itertools.product(acc[i], acc[j]):
            if na == nb and _overlap(ra, rb) and "w" in (ma, mb):
                fi, ki, si = nodes[i]
                fj, kj, sj = nodes[j]
                problems.append("frame %d step %d (op %d pass %d lane %d) and frame %d step %d (op %d pass %d lane %d) both touch %s rows %s / %s unordered"
                                % (fi, ki, si.op, si.pass_, si.lane, fj, kj, sj.op, sj.pass_, sj.lane, na, ra, rb))
                break
    return nodes, problems


@pytest.mark.parametrize("renderer,motion,strip_mode", [(api.RENDERER_BIASED, 0, 3), (api.RENDERER_BIASED, 24, 3), (api.RENDERER_BIASED, 0, 2), (api.RENDERER_BIASED, 24, 2),
                                                       (api.RENDERER_BIASED, 0, 1), (api.RENDERER_UNBIASED, 0, 3), (api.RENDERER_UNBIASED, 16, 3)])
def test_every_conflicting_access_of_three_frames_is_ordered(built_lib, renderer, motion, strip_mode):
    cfg = api.RestirRenderer.default_config(W, H, renderer)
    nodes, problems = _check(cfg, motion, strip_mode)
    assert not problems, "\n".join(problems[:8])
    lanes = {st.lane for _, _, st in nodes}
    assert {api.LANE_MAIN, api.LANE_GBUFFER, api.LANE_GATHER} <= lanes
    if renderer == api.RENDERER_BIASED and strip_mode == 2:
        assert api.LANE_SEAM in lanes          # the first of the two biased spatial passes runs its seam rows first
    if renderer == api.RENDERER_BIASED and strip_mode == 3:
        assert sum(1 for _, _, st in nodes if st.op == api.STEP_EXCHANGE_STRIPS and st.lane == api.LANE_MAIN) == 4 * (2 if motion else 1)   # one reservoir exchange per frame (+ the motion rows)


def test_the_checker_sees_a_missing_wait(built_lib, monkeypatch):
    """Drop the wait for the G-buffer strips (as if the exchange had been moved to the G-buffer lane and nothing else changed): the
    spatial pass that reads the neighbours' G-buffer rows is reported."""
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    real = api.frame_program

    def without_wait(*a):
        steps, r, b = real(*a)
        return [s for s in steps if s.op != api.STEP_WAIT_GBUFFER_STRIPS], r, b
    monkeypatch.setattr(api, "frame_program", without_wait)
    _, problems = _check(cfg, 0, 2)
    assert problems and any("gbuffer" in p for p in problems)


def test_recomputed_passes_leave_one_reservoir_exchange_per_frame(built_lib):
    """stripMode 3 (what the driver runs): the first of the two biased spatial passes runs on the band +- radius rows, fed by ONE exchange of
    radius x 2 rows of reservoirs + infos + pixel RNG states behind the candidate pass (G-buffer strips as tall, on the G-buffer lane); the
    last pass + shading run on the band; nothing is exchanged between the spatial passes."""
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    b, e = api.band_rows(H, WORLD, 3)
    cfg.rowBegin, cfg.rowEnd = b, e
    steps, _, _ = api.frame_program(cfg, 3, 0, False, 1, 0, False)
    xs = [s for s in steps if s.op == api.STEP_EXCHANGE_STRIPS]
    assert [(s.lane, s.exchangeRows, s.buffers) for s in xs] == [(api.LANE_GBUFFER, 2 * RADIUS, api.BUF_GBUFFERS), (api.LANE_MAIN, 2 * RADIUS, api.BUF_RESERVOIRS | api.BUF_RNG)]
    passes = [(s.pass_, s.rowBegin, s.rowEnd) for s in steps if s.op == api.STEP_RESTIR_PASS]
    assert passes == [(api.PASS_SETUP_GBUFFERS, b, e), (api.PASS_INITIAL_TEMPORAL_BIASED, b, e), (api.PASS_SPATIAL_BIASED, b - RADIUS, e + RADIUS),
                      (api.PASS_SPATIAL_BIASED_AND_SHADING, b, e)]
    order = [s.op for s in steps]
    assert order.index(api.STEP_WAIT_GBUFFER_STRIPS) < [i for i, s in enumerate(steps) if s.op == api.STEP_RESTIR_PASS and s.pass_ == api.PASS_SPATIAL_BIASED][0]
    # the unbiased estimator's single pass needs no halo: mode 3 is mode 1 there
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_UNBIASED)
    cfg.rowBegin, cfg.rowEnd = b, e
    a = [(s.op, s.pass_, s.rowBegin, s.rowEnd, s.exchangeRows, s.buffers, s.lane) for s in api.frame_program(cfg, 3, 0, False, 1, 0, True)[0]]
    c = [(s.op, s.pass_, s.rowBegin, s.rowEnd, s.exchangeRows, s.buffers, s.lane) for s in api.frame_program(cfg, 1, 0, False, 1, 0, True)[0]]
    assert a == c


def test_seam_rows_go_first_and_cover_what_the_neighbours_read(built_lib):
    cfg = api.RestirRenderer.default_config(W, H, api.RENDERER_BIASED)
    b, e = api.band_rows(H, WORLD, 3)
    cfg.rowBegin, cfg.rowEnd = b, e
    steps, _, _ = api.frame_program(cfg, 2, 0, False, 1, 0, False)
    ops = [(s.op, s.pass_, s.lane, s.rowBegin, s.rowEnd, s.gapBegin, s.gapEnd) for s in steps]
    k = next(i for i, s in enumerate(steps) if s.gapEnd > s.gapBegin)
    seam, xchg, interior = steps[k], steps[k + 1], steps[k + 2]
    assert (seam.rowBegin, seam.gapBegin, seam.gapEnd, seam.rowEnd) == (b, b + RADIUS, e - RADIUS, e), ops
    assert xchg.op == api.STEP_EXCHANGE_STRIPS and xchg.lane == api.LANE_SEAM and xchg.exchangeRows == RADIUS
    assert xchg.reservoirIndex == (seam.currentReservoirIndex + 1) % 2          # what the pass writes is what travels
    assert (interior.op, interior.pass_, interior.rowBegin, interior.rowEnd) == (api.STEP_RESTIR_PASS, api.PASS_SPATIAL_BIASED, b + RADIUS, e - RADIUS)
    assert interior.currentReservoirIndex == seam.currentReservoirIndex and interior.spatialNeighborBaseIndex == seam.spatialNeighborBaseIndex
    assert steps[k + 3].op == api.STEP_WAIT_SEAM_STRIPS
    # the top band has no seam above it, the bottom band none below
    for rank, want in ((0, (0, 0 + 0, None)), (WORLD - 1, None)):
        bb, ee = api.band_rows(H, WORLD, rank)
        cfg.rowBegin, cfg.rowEnd = bb, ee
        st = next(s for s in api.frame_program(cfg, 2, 0, False, 1, 0, False)[0] if s.gapEnd > s.gapBegin)
        assert (st.gapBegin, st.gapEnd) == ((bb, ee - RADIUS) if rank == 0 else (bb + RADIUS, ee))
    # stripMode 1 keeps every pass in one launch
    cfg.rowBegin, cfg.rowEnd = b, e
    assert not any(s.gapEnd > s.gapBegin or s.lane == api.LANE_SEAM for s in api.frame_program(cfg, 1, 0, False, 1, 0, False)[0])
