"""Randomised parity check of the ICP launch against the CPU oracle: patch sets of random sizes (empty, tiny, uneven,
beyond the LDS limits), random radii relative to the point spacing, georeferenced or local coordinates, both
estimators; float64 search must reproduce the oracle (<= 1e-9 m, equal iteration counts and correspondences)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fusion4landslide_amd import engine
from oracle import oracle as O

def rot(axis, ang):
    axis = np.asarray(axis, float); axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
# third argument "f32": the float32 search (fast mode) instead of the parity mode -- held to the fast mode's own bounds
# (tests/test_gpu_parity.py): a well-posed patch within 2e-3 m of the oracle, nine in ten within 1e-4 m
SEARCH = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] in ("f32", "f64") else "f64"
# "n32": float32 target normals for point-to-plane (f4l_patch_normals) instead of the doubles Open3D keeps (f4l_patch_normals_f64)
NORMALS_F64 = "n32" not in sys.argv[3:]
TRACE = "trace" in sys.argv[3:]  # (every mismatching patch again, pass by pass)
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.choice([1, 3, 40, 70, 600]))
    kind = rng.choice(["even", "uneven", "tiny", "one big"])
    if kind == "even": sizes = rng.integers(150, 400, P)
    elif kind == "uneven": sizes = rng.integers(0, 900, P)
    elif kind == "tiny": sizes = rng.integers(0, 12, P)
    else: sizes = np.r_[rng.integers(3000, 9500, 1), rng.integers(20, 200, max(P - 1, 0))]
    if sizes.sum() > 40000: sizes = (sizes * (40000 / sizes.sum())).astype(int)
    density = float(rng.choice([100.0, 400.0, 3000.0]))
    r = float(rng.choice([0.05, 0.1, 0.3]))
    origin = np.array([2647.0, 1177.0, 1500.0]) if rng.random() < 0.4 else np.zeros(3)  # (km-reduced: float32 keeps 0.1 mm there)
    icp_type = "point2plane" if rng.random() < 0.25 else "point2point"
    fixed = bool(rng.random() < 0.3)
    src_l, tgt_l = [], []
    for m in sizes:
        m = int(m)
        side = max(0.1, np.sqrt(max(m, 1) / density))
        mt = max(0, m + int(rng.integers(-5, 30))) if rng.random() < 0.9 else 0
        xy = rng.uniform(0, side, (mt, 2))
        t = np.c_[xy, 0.3 * np.sin(1.7 * xy[:, 0] / side) * np.cos(2.3 * xy[:, 1] / side) + rng.normal(0, 0.002, mt)]
        xy2 = rng.uniform(0.05 * side, 0.95 * side, (m, 2))
        s = np.c_[xy2, 0.3 * np.sin(1.7 * xy2[:, 0] / side) * np.cos(2.3 * xy2[:, 1] / side)]
        s = s @ rot(rng.normal(size=3), rng.uniform(0, 0.01)).T + rng.uniform(-0.4 * r, 0.4 * r, 3)
        src_l.append(s + origin); tgt_l.append(t + origin)
    src = np.concatenate(src_l).astype(np.float32) if len(src_l) else np.zeros((0, 3), np.float32)
    tgt = np.concatenate(tgt_l).astype(np.float32) if len(tgt_l) else np.zeros((0, 3), np.float32)
    soff = np.zeros(P + 1, np.int64); np.cumsum([len(a) for a in src_l], out=soff[1:])
    toff = np.zeros(P + 1, np.int64); np.cumsum([len(a) for a in tgt_l], out=toff[1:])
    if icp_type == "point2plane" and any(0 < len(a) < 3 for a in tgt_l):
        icp_type = "point2point"
    # which semantics of the point-to-plane step (include/f4l.h): Open3D's own against the strict restatement, or the robust default
    # against the oracle's robust variant (drawn from a generator of its own: the cases keep their seeds)
    sem = "open3d" if np.random.default_rng(seed0 + case + 7919).random() < 0.5 else "robust"
    otype = icp_type if icp_type != "point2plane" else ("point2plane" if sem == "open3d" else "point2plane_robust")
    t0 = time.perf_counter()
    ref = O.piecewise_icp(src, soff, tgt, toff, max_corr_dist=r, max_iter=30, icp_type=otype, fixed_iters=fixed)
    t1 = time.perf_counter()
    dv = lambda a: torch.from_numpy(a).cuda()
    nrm = engine.patch_normals(dv(tgt), dv(toff), 30, f64=NORMALS_F64) if icp_type == "point2plane" and len(tgt) else None  # (what the launch computes itself)
    out = engine.piecewise_icp(dv(src), dv(soff), dv(tgt), dv(toff), max_corr_dist=r, max_iter=30, icp_type=icp_type, p2plane=sem,
                               fixed_iters=fixed, search=SEARCH, tgt_normals=nrm)
    nrm_h = None if nrm is None else nrm.cpu().numpy().astype(np.float64)
    T = out["T"].cpu().numpy()
    fit = ref["fitness"]
    # (point-to-plane with float32 normals: 6e-8 relative on every normal, times what the patch's conditioning makes of it --
    #  1.6e-6 m on a 47-point patch whose radius exceeds its size, millimetres on a few; with the doubles Open3D keeps the
    #  kernel's normals and the oracle's differ in their eigen-solvers' last bits only: 1.1e-7 m was the most in 6000 sets)
    tol = 1e-9 if icp_type == "point2point" else (5e-7 if NORMALS_F64 else 5e-6)
    if SEARCH == "f32":
        tol = 2e-3
    worst, worst_posed, n_bad, n_bad_posed, n_unstable, n_order, n_normals, n_exit, n_posed, detail = 0.0, 0.0, 0, 0, 0, 0, 0, 0, 0, []
    it_k = out["iters"].cpu().numpy()
    for p in range(P):
        s = src[soff[p]:soff[p + 1]].astype(np.float64)
        if not len(s):
            continue
        e = float(np.abs((s @ T[p, :3, :3].T + T[p, :3, 3]) - (s @ ref["T"][p, :3, :3].T + ref["T"][p, :3, 3])).max())
        # a patch pins its six degrees of freedom only with enough well-spread correspondences; below that the two
        # sides may settle differently after the first rounding difference (that is chaos, not a defect)
        # (the fast mode is held to its bound only on patches that keep three quarters of their points matched: on a half-matched
        #  one -- a radius below the point spacing -- a float32 rounding moves a pair across the radius and the result by millimetres:
        #  2.6e-3 m at fitness 0.67, `fuzz_icp.py 1 7200213 f32`)
        posed = len(s) >= 40 and fit[p] >= (0.75 if SEARCH == "f32" else 0.5) and (toff[p + 1] - toff[p]) >= 40
        worst = max(worst, e)
        if e > tol and posed:
            # ... or with a trajectory that amplifies rounding by itself: the ORACLE, started a few ulps of the coordinates (1e-13 m at the
            # origin) away from the identity,
            # must land where it landed before -- if it does not, no second implementation can be held to it on this patch
            one = lambda a, off: np.ascontiguousarray(a[off[p]:off[p + 1]])
            z2 = np.array([0, len(s)], np.int64), np.array([0, int(toff[p + 1] - toff[p])], np.int64)
            Tp = np.eye(4)[None].copy()
            nudge = max(1e-13, 8 * 2.2e-16 * float(np.abs(s).max()))  # (a few ulps of the coordinates: less is rounded away)
            if SEARCH == "f32":
                nudge = 1e-6  # (the fast mode measures in float32 on patch-relative coordinates: a few of ITS ulps over a metre)
            Tp[0, :3, 3] = (nudge, -nudge, nudge)
            again = O.piecewise_icp(one(src, soff), z2[0], one(tgt, toff), z2[1], init_T=Tp, max_corr_dist=r, max_iter=30, icp_type=otype,
                                    fixed_iters=fixed)
            e_self = float(np.abs((s @ again["T"][0, :3, :3].T + again["T"][0, :3, 3]) - (s @ ref["T"][p, :3, :3].T + ref["T"][p, :3, 3])).max())
            if e_self > tol:
                posed = False
                n_unstable += 1
            else:
                # ... or on the ORDER of the sums: the oracle on the same patch with its source points in reverse order (Open3D
                # adds them up in whatever order its threads finish).  Point-to-plane far from the origin is the typical case:
                # the 6 x 6 system is solved in the caller's frame, its conditioning grows with (distance to the origin / patch
                # size)^2, the update is linearised about that origin, and the first pass differs by centimetres from the same
                # pass in the patch's own frame; on such a plateau the early exit, or a pair at the radius, turns on the last bits.
                sd, td = one(src, soff)[::-1].copy(), one(tgt, toff)
                rev = O.piecewise_icp(sd, z2[0], td, z2[1], max_corr_dist=r, max_iter=30, icp_type=otype, fixed_iters=fixed)
                e_rev = float(np.abs((s @ rev["T"][0, :3, :3].T + rev["T"][0, :3, 3]) - (s @ ref["T"][p, :3, :3].T + ref["T"][p, :3, 3])).max())
                if e_rev > tol:
                    posed = False
                    n_order += 1
                elif SEARCH == "f32":
                    # ... or, for the fast mode, on the float32 rounding of the patch-relative source coordinates, which is all a
                    # float32 search can see: the oracle on source points rounded that way must land where it landed before
                    s64, t64 = one(src, soff).astype(np.float64), one(tgt, toff).astype(np.float64)
                    o = t64[0] if len(t64) else np.zeros(3)
                    s_r = (s64 - o).astype(np.float32).astype(np.float64) * (1.0 + 2.0 ** -23) + o  # (one float32 step outward)
                    base = O.icp(s64, t64, max_corr_dist=r, max_iter=30, icp_type=otype, fixed_iters=fixed)
                    pert = O.icp(s_r, t64, max_corr_dist=r, max_iter=30, icp_type=otype, fixed_iters=fixed)
                    Tb, Tq = base["est_transform"], pert["est_transform"]
                    e_f32 = float(np.abs((s @ Tb[:3, :3].T + Tb[:3, 3]) - (s @ Tq[:3, :3].T + Tq[:3, 3])).max())
                    if e_f32 > tol:
                        posed = False
                        n_unstable += 1
                elif nrm_h is not None:
                    # ... or, point-to-plane, on the last bits of the NORMALS: the kernel is handed float32 normals, the oracle made
                    # its own in double.  The oracle run on the very normals the kernel had must land where the kernel landed.
                    same = O.icp(one(src, soff).astype(np.float64), one(tgt, toff).astype(np.float64), max_corr_dist=r, max_iter=30,
                                 icp_type=otype, fixed_iters=fixed, tgt_normals=np.ascontiguousarray(nrm_h[toff[p]:toff[p + 1]]))
                    Ts = same["est_transform"]
                    e_same = float(np.abs((s @ Ts[:3, :3].T + Ts[:3, 3]) - (s @ T[p, :3, :3].T + T[p, :3, 3])).max())
                    if e_same <= tol:
                        posed = False
                        n_normals += 1
                if posed and not fixed and int(it_k[p]) != int(ref["iters"][p]):
                    # ... or on the EXIT decision: both sides walk the same trajectory and leave it at different passes (the
                    # criteria compare a change of the rmse with 1e-6; on a plateau -- point-to-plane far from the origin -- the
                    # change crosses that value within rounding while a pass still moves the patch by more than the tolerance).
                    # The oracle made to run exactly the kernel's number of passes must land where the kernel landed.
                    forced = O.icp(one(src, soff).astype(np.float64), one(tgt, toff).astype(np.float64), max_corr_dist=r, max_iter=int(it_k[p]),
                                   icp_type=otype, fixed_iters=True,
                                   tgt_normals=None if nrm_h is None else np.ascontiguousarray(nrm_h[toff[p]:toff[p + 1]]))
                    Tf = forced["est_transform"]
                    e_forced = float(np.abs((s @ Tf[:3, :3].T + Tf[:3, 3]) - (s @ T[p, :3, :3].T + T[p, :3, 3])).max())
                    if e_forced <= tol:
                        posed = False
                        n_exit += 1
                        detail.append(("exit", p, int(it_k[p]), int(ref["iters"][p]), e, e_forced))
        if e > tol and posed and TRACE:
            # pass by pass: the kernel and the oracle on this patch alone, both made to run exactly k passes -- in the caller's frame
            # and with both clouds moved to the patch (exact in float32: the coordinates share the origin's exponent)
            one = lambda a, off: np.ascontiguousarray(a[off[p]:off[p + 1]])
            for frame, sh in (("caller's frame", np.zeros(3, np.float32)), ("patch frame", one(tgt, toff)[0])):
                s1, t1_ = one(src, soff) - sh, one(tgt, toff) - sh
                z = np.array([0, len(s1)], np.int64), np.array([0, len(t1_)], np.int64)
                n1 = None if nrm_h is None else np.ascontiguousarray(nrm_h[toff[p]:toff[p + 1]])
                line = []
                for kk in range(1, int(max(it_k[p], ref["iters"][p])) + 3):
                    o_ = O.icp(s1.astype(np.float64), t1_.astype(np.float64), max_corr_dist=r, max_iter=kk, icp_type=otype, fixed_iters=True, tgt_normals=n1)
                    k_ = engine.piecewise_icp(dv(s1), dv(z[0]), dv(t1_), dv(z[1]), max_corr_dist=r, max_iter=kk, icp_type=icp_type, p2plane=sem, fixed_iters=True,
                                              search=SEARCH, tgt_normals=None if n1 is None else dv(n1))
                    Tk, To = k_["T"].cpu().numpy()[0], o_["est_transform"]
                    s64 = s1.astype(np.float64)
                    line.append("%.1e (rmse %.9f)" % (float(np.abs((s64 @ Tk[:3, :3].T + Tk[:3, 3]) - (s64 @ To[:3, :3].T + To[:3, 3])).max()), o_["inlier_rmse"]))
                print(f"    patch {p}, {frame}: kernel against oracle after 1, 2, ... passes: " + ", ".join(line), flush=True)
        if e > tol:
            n_bad += 1
            if posed:
                n_bad_posed += 1
                detail.append((p, len(s), int(toff[p + 1] - toff[p]), round(float(fit[p]), 2), e, int(it_k[p]), int(ref["iters"][p])))
        if posed:
            worst_posed = max(worst_posed, e)
            n_posed += 1
    ok = n_bad_posed == 0
    exits = [d[1:] for d in detail if d[0] == "exit"]  # (patch, kernel passes, oracle passes, difference, difference at equal passes)
    if SEARCH == "f32":
        # the fast mode adds a float32 rounding to every pass: on a patch whose iteration does not settle (point-to-plane, a radius
        # beyond the patch, half the points matched: the oracle's rmse wanders for all 30 passes) that grows to centimetres where
        # the parity mode stays at 1e-13 m and a single nudge of the oracle's start shows little.  Held to what the test suite
        # holds it to: nearly all patches close (tests/test_gpu_parity.py), here no more than 2 % of a set's well-posed patches off
        ok = n_bad_posed <= n_posed // 50 + (1 if n_posed >= 30 else 0)
    bad += not ok
    print(f"case {seed0 + case:3d} P={P:4d} {kind:8s} n={len(src):6d} max_src={int(np.diff(soff).max()):5d} r={r} dens={density:6.0f} "
          f"{'geo' if origin[0] else 'loc'} {icp_type:11s} fixed={int(fixed)}  worst {worst:.1e} (well-posed patches {worst_posed:.1e}), "
          f"{n_bad} patches differ, {n_bad_posed} of them well-posed{f' ({n_unstable} more where the oracle itself moves by more than the tolerance when started 1e-13 m off)' if n_unstable else ''}"
          f"{f' ({n_order} more where the oracle moves by more than the tolerance when the source points come in reverse order)' if n_order else ''}"
          f"{f' ({n_normals} more where the oracle, given the float32 normals the kernel had, lands where the kernel landed)' if n_normals else ''}"
          f"{f' ({n_exit} more that left the same trajectory at another pass: the oracle run for exactly the passes of the kernel lands where the kernel landed {exits[:2]})' if n_exit else ''}  "
          f"{'ok' if ok else 'MISMATCH ' + str([d for d in detail if d[0] != 'exit'][:4])}", flush=True)
print("FUZZ", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
