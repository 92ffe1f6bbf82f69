"""Randomised differential test: random sparse SPD systems (sizes around every kernel-selection boundary, row lengths
from 1 to ~50, banded or scrambled) through every preconditioner kind, HIP path vs the C oracle.  Prints mismatches.

    python tools/fuzz_parity.py [cases] [seed] [only_case]
"""
import sys
import numpy as np
import scipy.sparse as sp
import torch
import deeppreconditioning_amd as D
from oracle import c_oracle as CO
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] != "chip" else None      # replay ONE case of a seed (same random draws), with its histories
CHIP_ONLY = "chip" in sys.argv[3:]       # `cases seed chip`: sizes, row lengths and bandwidths the whole-chip kernels take
SIZES = [1, 2, 63, 64, 255, 256, 257, 1000, 3071, 3072, 3073, 4608, 4609, 6144, 6145, 9000, 20000, 65536, 65537, 70000, 131073, 400000,
         600000, 1048576]        # (65 537 .. 1 048 576: the whole-chip kernels when rows and bandwidth allow)


def random_spd(n, per_row, band, scramble):
    k = max(1, per_row // 2)
    rows = np.repeat(np.arange(n), k)
    offs = rng.integers(1, max(2, band), size=n * k)
    cols = rows - offs
    keep = cols >= 0
    vals = rng.uniform(-1, 1, size=keep.sum())
    B = sp.coo_matrix((vals, (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    B.sum_duplicates()
    A = B + B.T
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + rng.uniform(0.1, 1.0, n))
    A = A.tocsr()
    if scramble:
        p = rng.permutation(n)
        A = A[p][:, p].tocsr()
    A.sort_indices()
    return A


bad = 0
chip_seen = {}
if CHIP_ONLY:
    SIZES = [6145, 9000, 20000, 65537, 70000, 100000, 131073, 262144, 262145, 400000, 524289, 600000, 1000000, 1048576]
for case in range(cases):
    n = int(rng.choice(SIZES))
    per_row = int(rng.choice([1, 2, 4, 6] if CHIP_ONLY else [1, 2, 4, 6, 10, 20, 50]))
    band = int(rng.choice([2, 8, 64, 1000, 20000] if CHIP_ONLY else [2, 8, 64, 1000, max(2, n)]))
    scramble = bool(rng.integers(0, 2)) and n < 100000
    if n * per_row > 6_000_000:
        per_row = 4
    A = random_spd(n, per_row, band, scramble)
    fp32_values = bool(rng.integers(0, 2))
    if fp32_values:                      # values that survive fp64 -> fp32 -> fp64: the lossless-fp32 mode must engage
        A.data = A.data.astype(np.float32).astype(np.float64)
    b = rng.uniform(-1, 1, n)
    x0 = rng.uniform(-1, 1, n) if rng.integers(0, 2) else None
    reorder = str(rng.choice(["auto", "auto", "rcm"]))          # a third of the cases force the library's reordering
    tag = (f"case {case}: n={n} nnz/row={A.nnz / n:.1f} band={band} scramble={scramble} fp32vals={fp32_values} "
           f"x0={x0 is not None} reorder={reorder}")
    if only is not None and case != only:      # consume what the case would have drawn, skip the work
        rng.uniform(-1, 1, n)
        if n <= 6145:
            rng.choice([0.0, 0.01, 0.1])
        continue
    S = D.CsrSystem.from_any(A, reorder=reorder)
    # a reordered handle iterates on B = P A P^T: that is the system the oracle is run on (vectors permuted alike); what the
    # caller hands over or gets back stays in the caller's numbering
    perm = S.permutation()
    if perm is None:
        perm = np.arange(n)
        B = A
    else:
        B = A[perm][:, perm].tocsr()
        B.sort_indices()
    bo = b[perm]
    x0o = None if x0 is None else x0[perm]
    x = rng.uniform(-1, 1, n)
    y = (S @ torch.from_numpy(x).cuda()).cpu().numpy()[perm]
    ref = CO.spmv(B, x[perm])
    kern = S.info()["spmv_kernel"]
    if kern == "vector":
        ok = np.allclose(y, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    else:
        ok = np.array_equal(y, ref)
    if not ok:
        bad += 1
        print("SPMV MISMATCH", tag, kern, np.abs(y - ref).max())
    # config 5's operator: fp32-stored values and vector, fp64 products and in-order sums, result rounded to fp32 once
    y32 = S.spmv_f32(torch.from_numpy(x.astype(np.float32)).cuda()).cpu().numpy()[perm]
    ref32 = CO.spmv_mixed(B, x[perm]).astype(np.float32)
    ok32 = (np.allclose(y32, ref32, rtol=1e-6, atol=1e-6 * np.abs(ref32).max()) if kern == "vector"
            else np.array_equal(y32, ref32))
    if not ok32:
        bad += 1
        print("SPMV-F32 MISMATCH", tag, kern, np.abs(y32 - ref32).max())
    # level-scheduled triangular solves on a factor with this matrix's structure: bit-exact whatever the segment forms
    Ltri = sp.tril(A, format="csr")
    Ltri.sort_indices()
    S.set_preconditioner(D.LLtSolve(Ltri))
    ylo = S.sptrsv(torch.from_numpy(x).cuda(), upper=False).cpu().numpy()
    ref_lo = CO.sptrsv_lower(Ltri, x)
    yup = S.sptrsv(torch.from_numpy(ref_lo).cuda(), upper=True).cpu().numpy()
    if not (np.array_equal(ylo, ref_lo) and np.array_equal(yup, CO.sptrsv_upper(CO.transpose_csr(Ltri), ref_lo))):
        bad += 1
        print("SPTRSV MISMATCH", tag, S.info()["levels_lower"])
    kinds = (["none", "jacobi"] + (["ic0_solve", "ic0_multiply"] if n <= 131073 else [])
             + (["ic0_csr", "ict_solve"] if n <= 6145 else []))
    Lf = None
    for kind in kinds:
        try:
            hist_np = None          # the numpy oracle: how far two correct implementations drift apart on this system
            if kind == "none":
                S.set_preconditioner(None)
                it, hist = CO.pcg(B, bo, "none", x0=x0o)[1:3]
                hist_np = np.array(O.preconditioned_conjugate_gradient(B, bo, O.Precond("none"), x0=x0o)[2])
            elif kind == "jacobi":
                S.set_preconditioner(D.Jacobi())
                it, hist = CO.pcg(B, bo, "jacobi", dinv=O.jacobi_dinv(B), x0=x0o)[1:3]
                hist_np = np.array(O.preconditioned_conjugate_gradient(B, bo, O.Precond("jacobi", dinv=O.jacobi_dinv(B)), x0=x0o)[2])
            elif kind == "ic0_solve":
                Lf = CO.ic0(A)
                S.set_preconditioner(D.IC0("solve"))
                it, hist = CO.pcg(A, b, "llt_solve", L=Lf, x0=x0)[1:3]
            elif kind == "ict_solve":    # level-1 fill + drop tolerance: device factor == CPU contract, then the solve
                thr = float(rng.choice([0.0, 0.01, 0.1]))
                Lt_ = O.ict(A, 1, thr)
                S.set_preconditioner(D.ICT("solve", 1, thr))
                frp, fci, fv = S.factor()
                if not (np.array_equal(frp, Lt_.indptr) and np.array_equal(fci, Lt_.indices) and np.array_equal(fv, Lt_.data)):
                    bad += 1
                    print("ICT FACTOR MISMATCH", tag, thr)
                it, hist = CO.pcg(A, b, "llt_solve", L=Lt_, x0=x0)[1:3]
            elif kind == "ic0_multiply":
                S.set_preconditioner(D.IC0("multiply"))
                it, hist = CO.pcg(A, b, "llt_multiply", L=Lf, x0=x0)[1:3]
            else:                        # M = L L^T as one explicit CSR: the reference's own technique (test.py:88)
                Mcsr = (Lf @ Lf.T).tocsr()
                Mcsr.sort_indices()
                S.set_preconditioner(Mcsr)
                it, hist = CO.pcg(A, b, "csr", M=Mcsr, x0=x0)[1:3]
            x0_dev = None if x0 is None else torch.from_numpy(x0).cuda()
            base = S.solve(torch.from_numpy(b).cuda(), x0_dev, flags=D._lib.NO_SMALL)
            if fp32_values:              # lossless fp32 value storage: bit-identical by construction
                v32 = S.solve(torch.from_numpy(b).cuda(), x0_dev, flags=D._lib.NO_SMALL | D._lib.VAL32_IF_LOSSLESS)
                if not (np.array_equal(v32.res_history, base.res_history) and torch.equal(v32.x, base.x)):
                    bad += 1
                    print("LOSSLESS-FP32 MISMATCH", tag, kind, v32.iterations, base.iterations)
            if kind in ("none", "jacobi", "ic0_solve"):
                # mixed precision (DPCG_SPMV_F32) against the mixed oracle.  Rounding p to fp32 is discontinuous, so two
                # correct implementations drift apart by far more than in fp64 (the two CPU oracles measure by how much
                # on this system): first entries tight, the rest within 30x that drift, count in a 2 % window.
                if kind == "ic0_solve":
                    kwm, Bm, bm, x0m = dict(L=Lf), A, b, x0
                else:
                    kwm, Bm, bm, x0m = (dict(dinv=O.jacobi_dinv(B)) if kind == "jacobi" else {}), B, bo, x0o
                okind = {"none": "none", "jacobi": "jacobi", "ic0_solve": "llt_solve"}[kind]
                itm, hm = CO.pcg(Bm, bm, okind, x0=x0m, mixed=True, **kwm)[1:3]
                hn = np.array(O.preconditioned_conjugate_gradient(O.MixedOperatorX0(Bm), bm, O.Precond(okind, **kwm), x0=x0m)[2])
                rm = S.solve(torch.from_numpy(b).cuda(), x0_dev, flags=D._lib.SPMV_F32 | D._lib.NO_SMALL)     # (the launches: the whole-chip kernel's mixed form is checked below)
                if kind in ("none", "jacobi"):
                    # round 4: the oracle sums in the DEVICE's reduction tree, so there is no drift to bound -- bit for bit
                    # (rows too long for the in-order kernels take the CSR-vector SpMV, whose shuffle-tree row sums are restated too)
                    itd, hd = CO.pcg(Bm, bm, okind, x0=x0m, mixed=True, device_tree=S.reduction_geometry(), **kwm)[1:3]
                    if not (rm.iterations == itd and np.array_equal(rm.res_history, hd)):
                        bad += 1
                        print("MIXED-PRECISION BITS MISMATCH", tag, kind, rm.iterations, itd)
                mm = min(len(hm), len(hn), len(rm.res_history))
                sigm = np.abs(hm[:mm]) > 1e-22
                drift_m = np.abs(hn[:mm] - hm[:mm])[sigm] / np.abs(hm[:mm])[sigm]
                rel_m = np.abs(rm.res_history[:mm] - hm[:mm])[sigm] / np.abs(hm[:mm])[sigm]
                tol_m = max(1e-5, 30 * float(drift_m.max()) if drift_m.size else 0.0)
                # ("first entries": the first quarter of the history, four at most -- a solve of four updates whose residual drops
                # 100x per update has no tight head: case 41 of seed 2024, HIP vs C 1.1e-8 where the two oracles differ by 6.2e-9)
                nh = max(1, min(4, mm // 4))
                head_ok = rel_m[:nh].size == 0 or float(rel_m[:nh].max()) < max(1e-9, 30 * float(drift_m[:nh].max()))
                # (the drift is chaotic: once two roundings of p have parted, the gap grows by a factor per update -- case 86 of seed
                # 477: equal to 1e-15 for nine updates, 2.4e-9 at update 15, x4 per update at the end, 1.9e-4 at update 36 where the two
                # CPU oracles are 2e-7 apart, counts equal.  So: the first half of the history within 30x the oracles' drift, the second
                # half within 1e-3.)
                half = rel_m.size // 2
                # (kept for IC(0) only, whose <r,z> the apply's last kernel sums in a tree the oracle does not restate; M = I and Jacobi
                # are held to the bits above)
                body_ok = (rel_m.size == 0 or (float(rel_m[:max(half, 1)].max()) < tol_m and float(rel_m.max()) < max(tol_m, 1e-3)))
                if not (head_ok and body_ok
                        and abs(rm.iterations - itm) <= 0.02 * itm + 1 + abs(len(hn) - 1 - itm)):
                    bad += 1
                    print("MIXED-PRECISION MISMATCH", tag, kind, "iters", rm.iterations, itm, len(hn) - 1, "max rel",
                          float(rel_m.max()) if rel_m.size else None, "oracle drift", float(drift_m.max()) if drift_m.size else None)
                if only is not None:
                    np.set_printoptions(precision=3, linewidth=200)
                    print(kind, "mixed GPU / C mixed oracle - 1:", rm.res_history[:mm] / hm[:mm] - 1)
                    print(kind, "mixed numpy oracle / C mixed oracle - 1:", hn[:mm] / hm[:mm] - 1)
                    print(kind, "C mixed history:", hm[:mm])
            tree_hist = None
            if kind in ("none", "jacobi"):     # round 4: the multi-launch forms against the device-tree oracle (CSR-vector kernel included), bit for bit
                tree_hist = CO.pcg(B, bo, kind, x0=x0o, device_tree=S.reduction_geometry(),
                                   **(dict(dinv=O.jacobi_dinv(B)) if kind == "jacobi" else {}))[1:3]
            # round 5: a plain call of a chip-sized system is ONE launch on the whole chip (dpcg_chip.hip / dpcg_chip_llt.hip): bit for
            # bit against the oracle with THAT tree (M = L L^T multiplied: on the matrices the handle iterates on)
            chip_hist = None
            ci = S.chip_info()
            if ci["chip_by_default"] and kind in ("none", "jacobi", "ic0_multiply"):
                ctree = {**S.reduction_geometry(), "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"], "lanes_per_row": ci["lanes_per_row"]}
                if kind == "ic0_multiply":
                    Lq = Lf[perm][:, perm].tocsr() if S.reordered else Lf
                    Lq.sort_indices()
                    chip_hist = CO.pcg(B, bo, "llt_multiply", L=Lq, x0=x0o, device_tree=ctree)[1:3]
                else:
                    chip_hist = CO.pcg(B, bo, kind, x0=x0o, device_tree=ctree, **(dict(dinv=O.jacobi_dinv(B)) if kind == "jacobi" else {}))[1:3]
                chip_seen[kind] = chip_seen.get(kind, 0) + 1
                # ... and config 5 in the same kernel (MODE 4; without a start vector: cg.py:60 reads the fp64 matrix)
                if kind in ("none", "jacobi") and x0_dev is None:
                    mixed_ref = CO.pcg(B, bo, kind, mixed=True, device_tree=ctree, **(dict(dinv=O.jacobi_dinv(B)) if kind == "jacobi" else {}))[1:3]
                    rm = S.solve(torch.from_numpy(b).cuda(), None, flags=D._lib.SPMV_F32)
                    chip_seen[kind + "_mixed"] = chip_seen.get(kind + "_mixed", 0) + 1
                    if not (rm.iterations == mixed_ref[0] and np.array_equal(rm.res_history, mixed_ref[1])):
                        bad += 1
                        print("BITS MISMATCH of the mixed whole-chip solve against the chip-tree oracle", tag, kind, rm.iterations, mixed_ref[0])
            for flags in (0, D._lib.NO_SMALL, D._lib.NO_SMALL | D._lib.NO_FUSE):
                r = S.solve(torch.from_numpy(b).cuda(), x0_dev, flags=flags)
                h = r.res_history
                if tree_hist is not None and flags and not (r.iterations == tree_hist[0] and np.array_equal(h, tree_hist[1])):
                    bad += 1
                    print("BITS MISMATCH against the device-tree oracle", tag, kind, "flags", flags, r.iterations, tree_hist[0])
                if chip_hist is not None and flags == 0 and not (r.iterations == chip_hist[0] and np.array_equal(h, chip_hist[1])):
                    bad += 1
                    print("BITS MISMATCH of the whole-chip solve against the chip-tree oracle", tag, kind, r.iterations, chip_hist[0])
                m = min(len(h), len(hist))
                # M = L L^T multiplied is the reference's own "unstable" technique (test.py:45): rounding differences
                # grow to O(1) within tens of updates, so only the first entries and a count window are comparable;
                # residuals at round-off level (an exact factorisation converges in one update) carry no information
                chaotic = kind in ("ic0_multiply", "ic0_csr")
                if hist_np is not None:   # the two oracles disagree with each other: no late history to compare against
                    mo = min(len(hist_np), len(hist))
                    big = np.abs(hist[:mo]) > 1e-22
                    d_all = np.abs(hist_np[:mo] - hist[:mo])[big] / np.abs(hist[:mo])[big]
                    chaotic = chaotic or len(hist_np) != len(hist) or (d_all.size and float(d_all.max()) > 1e-3)
                head = min(m, 8 if chaotic else m)
                if hist_np is not None and not chaotic and len(hist_np) == len(hist):
                    # superlinear end phase of a small system: rounding differences grow ~15x per update (seed 21, case 86: the
                    # two CPU oracles drift to 2e-6, the device to 6e-4 over the last nine updates).  Entries are comparable while
                    # the oracles still agree to 1e-13; the count is checked on all of them.
                    dd = np.abs(hist_np[:m] - hist[:m]) / np.maximum(np.abs(hist[:m]), 1e-300)
                    late = np.nonzero(dd > 1e-13)[0]
                    if late.size:
                        head = min(head, int(late[0]))
                sig = np.abs(hist[:head]) > 1e-22
                rel = np.abs(h[:head] - hist[:head])[sig] / np.abs(hist[:head])[sig]
                tol = 1e-6 if chaotic else 1e-8
                if S.reordered and kind not in ("none", "jacobi"):
                    tol = max(tol, 1e-7)     # factor-based M: the oracle runs in the caller's numbering, the sums differ in order
                if hist_np is not None and len(hist_np) == len(hist):     # rounding-order sensitivity of this system
                    drift = np.abs(hist_np[:head] - hist[:head])[sig] / np.abs(hist[:head])[sig]
                    tol = max(tol, 30 * float(drift.max()) if drift.size else tol)
                hist_ok = rel.size == 0 or float(rel.max()) < tol
                count_ok = abs(r.iterations - it) <= (0.06 * it + 2 if chaotic else 0)
                if only is not None:
                    np.set_printoptions(precision=6, linewidth=200)
                    print(kind, "flags", flags, "GPU / C oracle - 1:", (h[:m] / hist[:m] - 1))
                    if hist_np is not None:
                        print(kind, "numpy oracle / C oracle - 1:", (hist_np[:min(len(hist_np), len(hist))] / hist[:min(len(hist_np), len(hist))] - 1))
                if not (hist_ok and count_ok):
                    bad += 1
                    print("PCG MISMATCH", tag, kind, f"flags={flags}", "iters", r.iterations, it, "status", r.status,
                          "max rel err over the compared head", float(rel.max()) if rel.size else None)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("EXCEPTION", tag, kind, repr(e)[:200])
    S.close()
print(f"fuzz: {cases} cases, {bad} mismatches; whole-chip solves checked to the bit: {chip_seen}")
