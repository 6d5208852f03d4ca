"""Host-side models of two device schedules of gp_split256.hip's gemm_planes256_kernel, checked exhaustively on CPU:

* the work plan of a launch (data-parallel rounds + stream-K remainder per XCD chunk): every (tile, k-step) unit is
  computed exactly once, and every split tile's second part reads the fragment its first part publishes;
* the half-step-offset k loop of the two wave groups (one barrier per k-step): LDS slabs are complete before they are
  read and never overwritten while still being read.

The arithmetic below restates the kernel's index computations line by line (same names); the GPU tests check the
kernel itself against f64 and bit-for-bit against the lock-step kernel."""
import numpy as np
import pytest


def launch_plan(T, nstep, dp=True, slots=256):
    """-> {slot: [(tile, s0, s1, kind)]}, kind in {"dp", "head", "whole", "rest"}; tile is the global chunk-ordered index."""
    plan = {}
    for p in range(slots):
        x, n, slots_x = p & 7, p >> 3, slots >> 3
        t_lo = T * x // 8
        n_t = T * (x + 1) // 8 - t_lo
        rounds_dp = n_t // slots_x - 1 if (dp and n_t // slots_x > 1) else 0
        n_dp = rounds_dp * slots_x
        U = (n_t - n_dp) * nstep
        u0, u1 = U * n // slots_x, U * (n + 1) // slots_x
        ta, sa = divmod(u0, nstep)
        tb, sb = divmod(u1, nstep)
        n_head, n_rest = (1 if sb > 0 else 0), (1 if sa > 0 else 0)
        first_whole = ta + n_rest
        n_seg = rounds_dp + n_head + (tb - first_whole) + n_rest
        segs = []
        for seg in range(n_seg):
            is_dp = seg < rounds_dp
            is_head = (not is_dp) and seg - rounds_dp < n_head
            is_rest = (not is_dp) and n_rest == 1 and seg == n_seg - 1
            t = seg * slots_x + n if is_dp else n_dp + (tb if is_head else (ta if is_rest else first_whole + seg - rounds_dp - n_head))
            assert 0 <= t < n_t and not (is_head and is_rest)
            s0, s1 = (sa if is_rest else 0), (sb if is_head else nstep)
            segs.append((t_lo + t, s0, s1, "dp" if is_dp else "head" if is_head else "rest" if is_rest else "whole"))
        plan[p] = segs
    return plan


@pytest.mark.parametrize("T,nstep", [(1040, 32), (520, 32), (260, 32), (260, 128), (256, 2), (272, 1), (1056, 2), (2080, 4),
                                      (300, 3), (4096, 1), (257, 7)])
@pytest.mark.parametrize("dp", [True, False])
def test_launch_plan_covers_every_unit_once_and_pairs_hand_offs(T, nstep, dp):
    plan = launch_plan(T, nstep, dp)
    seen = {}
    for p, segs in plan.items():
        kinds = [k for *_, k in segs]
        # order inside a slot: data-parallel rounds, then the published head, whole tiles, the received rest last
        assert kinds == sorted(kinds, key=["dp", "head", "whole", "rest"].index)
        for tile, s0, s1, _ in segs:
            assert s0 < s1
            for s in range(s0, s1):
                assert (tile, s) not in seen, (tile, s)
                seen[(tile, s)] = p
    assert len(seen) == T * nstep
    heads = {p: (t, s1) for p, segs in plan.items() for t, s0, s1, k in segs if k == "head"}
    for p, segs in plan.items():
        for t, s0, s1, k in segs:
            if k == "rest":  # continues the tile exactly where slot p - 8 (same XCD, previous slot) stopped
                assert heads.get(p - 8) == (t, s0), (p, t, s0)
            if k == "head":
                assert s0 == 0   # the chain of a split tile starts at k = 0 in the publishing slot
    if dp:  # the data-parallel part puts the 32 slots of an XCD on 32 consecutive tiles of its chunk
        for p, segs in plan.items():
            for r, (t, s0, s1, k) in enumerate(segs):
                if k == "dp":
                    x, n = p & 7, p >> 3
                    assert t == T * x // 8 + r * 32 + n and (s0, s1) == (0, nstep)


def group_events(grp, ns):
    """The k loop of one wave group: C(s) = matrix phase on slab s, M(s) = stage slab s (+ load s+1), B = workgroup barrier."""
    ev = []
    if grp:
        ev.append(("M", 1))
    for s in range(ns):
        ev.append(("C", s))
        if grp and s + 1 < ns:
            ev.append("B")
        ev.append(("M", s + 1 + grp))
        if (not grp) and s + 1 < ns:
            ev.append("B")
    return ev


@pytest.mark.parametrize("ns", list(range(1, 12)) + [32, 128])
def test_half_step_offset_loop_has_no_lds_hazard(ns):
    ev = [group_events(0, ns), group_events(1, ns)]
    assert ev[0].count("B") == ev[1].count("B") == ns - 1          # same barrier count: no deadlock
    epoch = [{}, {}]                                               # (kind, slab) -> index of the barrier interval
    for g in (0, 1):
        k = 0
        for e in ev[g]:
            if e == "B":
                k += 1
            elif e[0] == "C" or e[1] < ns:                         # M of a slab >= ns stages nothing
                epoch[g][e] = k
    for slab in range(1, ns):
        for g in (0, 1):
            for gg in (0, 1):
                # slab complete: every group's stage of it lies in an EARLIER barrier interval than any read of it
                # (or the same interval of the same wave, which program order covers -- never the case here)
                assert epoch[gg][("M", slab)] < epoch[g][("C", slab)], (ns, slab, g, gg)
                # buffer reuse: slab (same buffer as slab - 2) is staged only after every read of slab - 2
                if slab >= 2:
                    assert epoch[g][("M", slab)] > epoch[gg][("C", slab - 2)], (ns, slab, g, gg)
    # the offset: group 1 stages while group 0 computes, in the first half of every interval
    assert ev[0][0] == ("C", 0) and ev[1][0] == ("M", 1)


# ---------------------------------------------------------------------------------------------------------------
# strip_grab (gp_split256.hip): the ragged strip's fragments are handed out on demand through ONE word per scratch,
# [launch tag : 20 | count : 12], never reset between launches.  Host model of the protocol at the granularity of its atomic
# operations (load / fetch-add / compare-and-swap), run under random interleavings: every fragment goes to exactly one slot,
# whatever the word held before (zeros, the previous launch's tag and count, garbage).
def _grab(word, tag):
    """Generator: yields once per atomic operation (the scheduler may switch slots there); returns the fragment index."""
    while True:
        cur = word[0]                                   # atomic load
        yield
        if (cur & ~0xfff) == tag:
            old = word[0]                               # fetch-add
            word[0] = old + 1
            yield
            return old & 0xfff
        ok = word[0] == cur                             # compare-and-swap(expected = cur, desired = tag | 1)
        if ok:
            word[0] = tag | 1
        yield
        if ok:
            return 0


def _slot(word, tag, nfrag, got):
    while True:
        f = yield from _grab(word, tag)
        if f >= nfrag:
            return
        got.append(f)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("nslots,nfrag", [(256, 64), (256, 128), (8, 1), (3, 40)])
def test_strip_fragments_are_handed_out_exactly_once(seed, nslots, nfrag):
    rs = np.random.RandomState(seed)
    word = [0]
    for launch in range(4):                             # consecutive launches on the same scratch: tags differ, nothing is reset
        if launch == 2:
            word[0] = int(rs.randint(0, 2 ** 31))       # and an arbitrary left-over
        tag = ((0x40000000 | (7 + launch)) & 0x7ffff) << 12
        got = [[] for _ in range(nslots)]
        live = {i: _slot(word, tag, nfrag, got[i]) for i in range(nslots)}
        order = list(live)
        while live:
            i = order[rs.randint(len(order))]
            try:
                next(live[i])
            except StopIteration:
                del live[i]
                order.remove(i)
        allf = sorted(f for g in got for f in g)
        assert allf == list(range(nfrag))
        assert (word[0] & ~0xfff) == tag and (word[0] & 0xfff) == nfrag + nslots   # every slot overshoots exactly once


# ---- the parallel split-K plan (fewer tiles than slots: gemm_planes256_kernel<.., PAR = true>) with the host's cap of at least
# kParMinSteps k-steps per slot of a split tile (gp_gemm_planes256_launch: a.par = max(1, nstep / kParMinSteps)).
def par_plan(T, nstep, min_steps=16, slots=256):
    """-> ({slot: (tile, s0, s1, part, S)} for slots that hold a tile, S per XCD chunk); tile is the global chunk-ordered index."""
    cap = max(1, nstep // min_steps)
    plan, S_of = {}, {}
    for p in range(slots):
        x, n, slots_x = p & 7, p >> 3, slots >> 3
        t_lo = T * x // 8
        n_t = T * (x + 1) // 8 - t_lo
        par_S = max(1, min(cap, slots_x // max(n_t, 1)))
        S_of[x] = par_S
        tile, part = n // par_S, n % par_S
        if tile < n_t:
            u0 = tile * nstep + nstep * part // par_S
            u1 = tile * nstep + nstep * (part + 1) // par_S
            assert u0 // nstep == tile and (u1 - 1) // nstep == tile
            plan[p] = (t_lo + tile, u0 - tile * nstep, u1 - tile * nstep, part, par_S)
    return plan, S_of


@pytest.mark.parametrize("T,nstep", [(64, 32), (64, 128), (128, 32), (192, 32), (32, 32), (32, 128), (8, 32), (96, 32), (255, 32), (40, 4),
                                      (24, 2048), (100, 48)])
@pytest.mark.parametrize("min_steps", [1, 16])
def test_parallel_split_plan_covers_every_unit_once_with_the_owner_last(T, nstep, min_steps):
    plan, S_of = par_plan(T, nstep, min_steps)
    seen, parts = {}, {}
    for p, (tile, s0, s1, part, S) in plan.items():
        assert s0 < s1, "a slot of a split tile without k-steps"
        for s in range(s0, s1):
            assert (tile, s) not in seen
            seen[(tile, s)] = p
        parts.setdefault(tile, []).append((part, p, s0, s1, S))
    assert len(seen) == T * nstep
    for tile, ps in parts.items():
        ps.sort()
        S = ps[0][4]
        assert [q[0] for q in ps] == list(range(S))                       # parts 0 .. S - 1, the owner (epilogue) holds the LAST k range
        assert ps[-1][3] == nstep and ps[0][2] == 0
        assert all(b[1] == a[1] + 8 for a, b in zip(ps, ps[1:]))          # neighbouring slots of ONE XCD (flags / partials of slot p - 8 m)
        if min_steps > 1 and S > 1:
            assert min(q[3] - q[2] for q in ps) >= min(min_steps, nstep // S), "a split tile's slot fell below the k-step floor"
    if min_steps == 16 and nstep == 32:
        assert max(S_of.values()) <= 2                                     # K = 1024: at most two slots per tile (proj / v below 64 crops)


# ---- the tile-shape policy of gp_gemm_planes256_launch below 64 crops (round 4): 256 x 128 tiles (NJ = 2) for epilogues 3 / 6 / 7 when the
# 256 x 256 tiles fill at most half the 256 slots; the parallel split then runs on twice the tiles with half-size partial accumulators.
def vit_large_layer_plans(crops, half_tiles=True):
    """-> {gemm: (tile columns, tiles, S, slots busy, partial MB moved)} for q|k|v, proj, fc1, fc2 of a ViT-L layer at `crops` crops."""
    j_main = (257 * crops) // 256 * 256
    out = {}
    for name, I, K, can_split in (("qkv", 3072, 1024, True), ("proj", 1024, 1024, True), ("fc1", 4096, 1024, False), ("fc2", 1024, 4096, True)):
        t256 = (I // 256) * (j_main // 256)
        if t256 >= 256:
            out[name] = (256, t256, 1, 256, 0.0)
            continue
        half = half_tiles and 2 * t256 <= 256
        T, width = (2 * t256, 128) if half else (t256, 256)
        # the GELU build keeps whole tiles (S = 1); so does (round 5) every 256-wide launch of 129-255 tiles: the launcher takes the
        # PSPLIT = false instantiation for them (an XCD holding exactly 16 of 132 tiles would otherwise have split them two ways)
        whole = (not can_split) or (width == 256 and T > 128)
        plan, S_of = par_plan(T, K // 32, 10 ** 9 if whole else 16)
        S = max(S_of.values())
        partial_mb = sum(1 for (_, _, _, part, s) in plan.values() if part < s - 1) * 256 * width * 4 / 1e6
        out[name] = (width, T, S, len(plan), partial_mb)
    return out


def test_half_width_tile_policy_at_eight_and_sixteen_crops():
    """The launch plans DESIGN.md section 4 (round 4) quotes: at 8 crops every GEMM of a layer takes the 256 x 128 tiles -- q|k|v and fc1 become
    whole tiles, proj two slots per tile on half the chip, fc2 four slots per tile with 25 MB of partials instead of 59 MB; at 16 crops only
    proj / fc2 do; from 32 crops on nothing changes for q|k|v / fc1 and proj / fc2 fill the chip with whole half-width tiles."""
    new, old = vit_large_layer_plans(8), vit_large_layer_plans(8, half_tiles=False)
    assert new["qkv"][:4] == (128, 192, 1, 192) and old["qkv"][:4] == (256, 96, 2, 192)
    assert new["fc1"][:4] == (128, 256, 1, 256) and old["fc1"][:4] == (256, 128, 1, 128)
    assert new["proj"][:4] == (128, 64, 2, 128) and old["proj"][:4] == (256, 32, 2, 64)
    assert new["fc2"][:4] == (128, 64, 4, 256) and old["fc2"][:4] == (256, 32, 8, 256)
    assert abs(new["fc2"][4] - 25.2) < 0.1 and abs(old["fc2"][4] - 58.7) < 0.1
    p16 = vit_large_layer_plans(16)
    assert p16["qkv"][:2] == (256, 192) and p16["fc1"][:2] == (256, 256) and p16["proj"][:4] == (128, 128, 2, 256) and p16["fc2"][:4] == (128, 128, 2, 256)
    p32 = vit_large_layer_plans(32)
    assert p32["qkv"][0] == 256 and p32["fc1"][0] == 256 and p32["proj"][:4] == (128, 256, 1, 256) and p32["fc2"][:4] == (128, 256, 1, 256)
    assert all(v[0] == 256 for v in vit_large_layer_plans(64).values())


@pytest.mark.parametrize("crops", [4, 8, 12, 16, 24, 32, 48])
def test_half_width_tile_plans_cover_every_unit(crops):
    for name, (width, T, S, busy, _) in vit_large_layer_plans(crops).items():
        if T >= 256:
            continue
        nstep = 128 if name == "fc2" else 32
        plan, _ = par_plan(T, nstep, 16 if name != "fc1" else 10 ** 9)
        assert sum(s1 - s0 for (_, s0, s1, _, _) in plan.values()) == T * nstep and busy == len(plan) <= 256


def test_full_width_parallel_builds_only_ever_run_whole_tiles():
    """Round 5 (profiles/r05_b16_regression.txt): with the 256 x 128 tiles taking every launch of at most 128 tiles, a 256 x 256 tile launch
    below 256 tiles holds 129-255 tiles, and the launcher runs them as ONE whole tile per slot (S = 1, no partial accumulators) on the
    PSPLIT = false instantiations (no split-K reduction compiled in: 235 / 251 VGPRs, no scratch) -- q|k|v at 11-21 crops, proj / fc2 at
    33-63.  (Only an XCD holding exactly 16 tiles -- 132 tiles: 11 / 33 crops -- would have split two ways under the old rule.)"""
    seen = set()
    for crops in range(4, 64):
        for name, (width, T, S, busy, partial_mb) in vit_large_layer_plans(crops).items():
            if width == 256 and T < 256:
                assert 128 < T < 256 and S == 1 and busy == T and partial_mb == 0.0, (crops, name, T, S)
                seen.add(name)
    assert {"qkv", "proj", "fc2"} <= seen
