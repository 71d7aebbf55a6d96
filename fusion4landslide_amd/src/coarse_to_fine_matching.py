"""Counterpart of the reference's `Coarse2Fine(cfg).implement_c2f_matching()` (src/coarse_to_fine_matching.py:195-290 over
src/coarse_to_fine_matching_base.py) for the 3D hot path of a tile: same cfg keys, same stage names, same result files.

    _read_data                             base:890-912     the two tile PLYs -> float32 (N, 3) on the device
    _voxel_subsampling                     base:1012-1057   engine.voxel_subsampling (median resolution, voxel grid, voxel <-> point maps)
    implement_partition                    base:2658-2694   computeSupervoxel x 2 at max(sqrt(3) * 10 * median resolution, voxel size)
                                                            (0.1 for rockfall_simulator), partition text files like the reference's
    load_partition / prepare_pts2spt_dict  base:1237-1332   labels -> CSR patches (f4l_labels_to_csr)
    global_matches_from_3d                 base:2756-2889   POINT MATCHES -- the reference finds them in a learned feature space (DIP
                                                            descriptors + faiss / hnsw, out of scope): supplied by the caller
                                                            (cfg.point_matches_3d, an (N_src,) index array, or cfg.point_matches_3d_fn),
                                                            else the stand-in: nearest target point within `parameter_setting.max_magnitude`
    (global matches from 2D)               base:1670-1675   `corres_3d_from_2d_idx` -- the product of the out-of-scope image pipeline:
                                                            supplied by the caller (cfg.point_matches_from_2d) or absent
    coarse_matching_with_different_types   base:2925-3233   PATCH MATCHES -- learned aggregated features in the reference: supplied by the
                                                            caller (cfg.patch_matches_fn), else the stand-in: a source patch is matched
                                                            with the target patch most of its point matches lead to
    fine_matching_with_different_types     base:3236-3436   src/fine_matching.fine_matching_3d (one batched call instead of the Python loop)
    save_process_dvf                       base:3459-3600   the c2f_* result files ('%.6f', the two visualisation rows)

What is NOT here: image matching and lifting, learned descriptors and aggregation, the superpoint partition (the reference's yaml
default `partition_type: superpoint` needs the absent superpoint_transformer submodule: run with `partition_type: supervoxel`).
"""
import ctypes as C
import os
import os.path as osp

import numpy as np

from .. import engine
from .._lib import check, lib
from ..cpp_core.supervoxel_segmentation.build import supervoxel as supervoxel_partition
from ..utils import async_io
from ..utils.common import AttrDict, dir_exist
from ..utils.ply import read_xyz32
from .fine_matching import fine_matching_3d, fine_matching_finish, fine_matching_prepare  # noqa: F401


def _get(d, key, default=None):
    return d[key] if (d is not None and key in d and d[key] is not None) else default


def _write_rows(path, rows):
    check(lib().f4l_write_rows_txt(path.encode(), rows.ctypes.data_as(C.c_void_p), rows.shape[0], rows.shape[1]), "f4l_write_rows_txt")


def max_mag_visualize(dataset):
    """The upper end of the colour scale the reference plants into row 1 of every *_visualize file (base:3480-3489)."""
    return {"rockfall_simulator": 0.06, "brienz_tls": 5, "mattertal": 10}.get(dataset, 10)


def save_process_dvf(output_root, tile_id, dataset, dense, sparse=None, tgt2src=None, multiple_case=True, voxel_size=None, defer=False):
    """`save_process_dvf` (src/coarse_to_fine_matching_base.py:3459-3600): the result files of a tile under <output_root>/results,
    with the reference's names, '%.6f', and -- in the *_visualize files -- magnitudes 0 and max_mag_visualize planted into rows 0
    and 1.  dense / sparse / tgt2src: (m, 6) arrays or tensors [from xyz, to xyz].  Returns the list of files written -- with
    `defer`, of files BEING written by the writer threads (utils/async_io; the tile loop drains them before the run returns)."""
    def host(a):
        if a is None:
            return None
        a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
        return a.astype(np.float32, copy=False)

    dense, sparse, tgt2src = host(dense), host(sparse), host(tgt2src)
    res_dir = osp.join(output_root, "results")
    dir_exist(res_dir)
    cap = max_mag_visualize(dataset) if multiple_case else (0.06 if dataset == "rockfall_simulator" else 5)
    written = []

    def save(name, arr):
        # np.savetxt(path, arr, delimiter=" ", fmt="%.6f") byte for byte (f4l_write_rows_txt: numpy formats value by value in
        # Python -- 2.4 s per million rows of six, four to eight such files per tile, against 30 ms of device work for the tile).
        # `arr`: the table, or a function that makes it (the magnitude tables: 30 ms of numpy per million rows, which a deferred
        # file leaves to its writer thread)
        path = osp.join(res_dir, name)

        def job():
            rows = np.ascontiguousarray(arr() if callable(arr) else arr, dtype=np.float32)
            if rows.ndim != 2:
                raise ValueError("save_process_dvf writes tables")
            _write_rows(path, rows)
        if defer:  # (the inputs are this call's own arrays: nobody touches them until the writer is through)
            async_io.submit(job)
        else:
            job()
        written.append(path)

    def xyz_mag(rows, planted):
        def make():
            mag = np.linalg.norm(rows[:, 3:6] - rows[:, :3], axis=1).astype(np.float32)[:, None]  # torch.linalg.norm on float32 (:3463-3465)
            if planted:
                mag = mag.copy()
                mag[0] = 0
                mag[1] = cap
            return np.hstack((rows[:, :3], mag))
        return make

    if multiple_case:
        save(f"c2f_dense_dvfs_src2tgt_tile_{tile_id}.txt", dense)                                   # :3477-3479
        save(f"c2f_dense_dvfms_src2tgt_tile_{tile_id}.txt", xyz_mag(dense, False))                  # :3480-3484
        # (the reference plants the two values into the SAME tensor it then keeps using: nothing else reads it afterwards)
        save(f"c2f_dense_dvfms_src2tgt_visualize_tile_{tile_id}.txt", xyz_mag(dense, True))         # :3498-3504
        if sparse is not None and len(sparse):
            save(f"c2f_sparse_dvfms_src2tgt_visualize_tile_{tile_id}.txt", xyz_mag(sparse, True))   # :3507-3516
        if tgt2src is not None:
            save(f"c2f_dense_dvfms_tgt2src_tile_{tile_id}.txt", xyz_mag(tgt2src, False))            # :3522-3526
            save(f"c2f_dense_dvfms_tgt2src_visualize_tile_{tile_id}.txt", xyz_mag(tgt2src, True))   # :3528-3537
            if voxel_size is not None and len(tgt2src):
                # the tgt2src rows whose start lies within a voxel of a src2tgt row's start, and the rest (:3539-3568; the
                # reference queries a cKDTree of the tgt2src starts with the src2tgt starts: f4l_nn_query here)
                import torch
                t2s = torch.from_numpy(np.ascontiguousarray(tgt2src[:, :3])).cuda()
                s2t = torch.from_numpy(np.ascontiguousarray(dense[:, :3])).cuda()
                idx, d2 = engine.nn_query(t2s, s2t, 1, return_d2=True)
                hit = (torch.sqrt(d2[:, 0]) < float(voxel_size)).cpu().numpy()
                idx = idx[:, 0].cpu().numpy()
                remain = np.ones(len(tgt2src), dtype=bool)
                remain[idx[hit]] = False
                vis = xyz_mag(tgt2src, True)()  # (the reference indexes the already planted magnitudes, then plants again)
                for name, sel in ((f"c2f_dvfms_tgt2src_mutual_intersect_with_src2tgt_visualize_tile_{tile_id}.txt", idx[hit]),
                                  (f"c2f_dvfms_tgt2src_mutual_remain_with_src2tgt_visualize_tile_{tile_id}.txt", np.nonzero(remain)[0])):
                    part = vis[sel].copy()
                    if len(part) > 1:
                        part[0, 3], part[1, 3] = 0, cap
                    save(name, part)
    else:
        save("c2f_dvfs_src2tgt.txt", dense)                                                         # :3570-3571
        save("c2f_dvfms_src2tgt.txt", xyz_mag(dense, False))
        save("c2f_dvfms_src2tgt_visualize_0_5.txt", xyz_mag(dense, True))
        if sparse is not None and len(sparse):
            save("c2f_dvfms_src2tgt_discrete_visualize_0_5.txt", xyz_mag(sparse, True))
    return written


class Coarse2Fine:
    """`Coarse2Fine(cfg)` of src/coarse_to_fine_matching.py:195 for `partition_type: supervoxel`.  cfg is the reference's nested
    config (main_fusion.py:63-78) with `tile_id`, `src_tile_overlap_path`, `tgt_tile_overlap_path` set by the tile loop."""

    def __init__(self, config):
        self.config = config
        self.method = config.method
        self.para = config.parameter_setting
        self.data = config.data
        self.verbose = bool(_get(config, "verbose", False))
        self.logging = config.logging
        self.output_root = config.path_name.output_root
        self.device = _get(config, "device")
        self.tile_id = _get(config, "tile_id", 0)  # (captured now: the tile loop moves the config on before a batched tile is finished)
        self.defer_files = bool(_get(config, "defer_files", False))  # (the tile loop's: partition / result files by the writer threads)
        self.data_input_3d, self.data_interim, self.data_output = AttrDict(), AttrDict(), AttrDict()
        if self.method.partition_type != "supervoxel":
            raise NotImplementedError(f"partition_type {self.method.partition_type!r}: only 'supervoxel' is built here (the superpoint "
                                      "partition wraps the absent superpoint_transformer submodule)")
        self.matching = ("only_3d" if _get(self.method, "fine_matching_only_3d", False) else
                         "only_2d" if _get(self.method, "fine_matching_only_2d", False) else
                         "fusion" if _get(self.method, "fine_matching_fusion", False) else None)
        if self.matching is None:
            raise NotImplementedError  # base:3275-3276
        self._read_data()

    # ---- base:890-912 -------------------------------------------------------------------------------------------------------
    def _read_data(self):
        import torch
        if self.data.multiple_case:
            self.src_pcd_path, self.tgt_pcd_path = self.config.src_tile_overlap_path, self.config.tgt_tile_overlap_path
        else:
            root = self.config.path_name.input_root
            self.src_pcd_path, self.tgt_pcd_path = osp.join(root, "raw_pcd", self.data.src_pcd), osp.join(root, "raw_pcd", self.data.tgt_pcd)
        dev = torch.device("cuda", torch.cuda.current_device())
        # (the tile loop may have started these reads while the tile before was on the device: utils/tiles.py, utils/async_io.py)
        self._host_xyz = {"src": async_io.take(self.src_pcd_path, read_xyz32), "tgt": async_io.take(self.tgt_pcd_path, read_xyz32)}
        self.data_input_3d.src_pts = torch.from_numpy(self._host_xyz["src"]).to(dev)
        self.data_input_3d.tgt_pts = torch.from_numpy(self._host_xyz["tgt"]).to(dev)

    # ---- base:1012-1057 -----------------------------------------------------------------------------------------------------
    def _voxel_subsampling(self):
        sub = engine.voxel_subsampling(self.data_input_3d.src_pts, self.data_input_3d.tgt_pts)
        self.method.voxel_size = sub["voxel_size"]          # `self.method.voxel_size = self._compute_median_resolution()`
        self.para.median_max_resolution = sub["voxel_size"]  # base:2749-2750
        self.data_interim.src_pts_sub, self.data_interim.tgt_pts_sub = sub["src"]["pts_sub"], sub["tgt"]["pts_sub"]
        self.data_interim.idx_voxel2pts_src, self.data_interim.idx_voxel2pts_tgt = sub["src"]["idx_voxel2pts"], sub["tgt"]["idx_voxel2pts"]
        self.data_interim.idx_pts2voxel_src, self.data_interim.idx_pts2voxel_tgt = sub["src"]["idx_pts2voxel"], sub["tgt"]["idx_pts2voxel"]

    def _compute_median_resolution(self):
        """base:2716-2754 on the SUBSAMPLED clouds, as `implement_partition` calls it after `_voxel_subsampling`."""
        res = engine.median_resolution(self.data_interim.src_pts_sub, self.data_interim.tgt_pts_sub)
        self.para.median_max_resolution = res
        return res

    # ---- base:2658-2694, 1237-1332 ---------------------------------------------------------------------------------------------
    def implement_partition(self):
        import torch
        svl_radius = max(np.sqrt(3) * (10 * self._compute_median_resolution()), self.method.voxel_size)
        if self.data.dataset == "rockfall_simulator":
            svl_radius = 0.1
        self.data_interim.svl_radius = float(svl_radius)
        partition_path = osp.join(self.output_root, f"{self.method.partition_type}_partition")
        dir_exist(partition_path)
        tag = f"_tile_{self.tile_id}" if self.data.multiple_case else ""
        save = bool(_get(self.method, "save_partition", True))
        labels = []
        # `computeSupervoxel(path, k, radius, out)` of base:2680-2694 without its second read of the same PLY and without the labels'
        # trip to the host and back: the cloud `_read_data` put on the device, the partition file by the writer threads
        for which, pts in (("src", self.data_input_3d.src_pts), ("tgt", self.data_input_3d.tgt_pts)):
            out = osp.join(partition_path, f"partition_of_input_{which}{tag}.txt") if save else "None"
            lab, _ = supervoxel_partition.computeSupervoxelDevice(pts, int(self.para.n_normals), float(svl_radius), out,
                                                                  xyz_host=self._host_xyz.get(which), defer=self.defer_files)
            labels.append(lab)
        self._host_xyz = {}
        self.data_interim.idx_pts2spt_src, self.data_interim.idx_pts2spt_tgt = labels

    def load_partition(self):
        """The reference re-reads column 6 of the partition text (base:1257-1276); the labels are still on the device here."""

    def prepare_pts2spt_dict(self):
        """base:1301-1332 (a boolean mask per label, O(K N)) as one sort by label: CSR patches of both epochs."""
        for which in ("src", "tgt"):
            lab = self.data_interim[f"idx_pts2spt_{which}"]
            K = int(lab.max().item()) + 1 if lab.numel() else 0
            order, off = engine.labels_to_csr(lab, K)
            self.data_interim[f"spt_order_{which}"], self.data_interim[f"spt_off_{which}"] = order.to("cuda").long(), off

    # ---- base:2756-2889 (stand-in / hook) --------------------------------------------------------------------------------------
    def global_matches_from_3d(self):
        import torch
        src, tgt = self.data_input_3d.src_pts, self.data_input_3d.tgt_pts
        given = _get(self.config, "point_matches_3d")
        fn = _get(self.config, "point_matches_3d_fn")
        if given is not None:
            corr = torch.as_tensor(given, dtype=torch.int64, device=src.device)
        elif fn is not None:
            corr = torch.as_tensor(fn(src, tgt), dtype=torch.int64, device=src.device)
        else:  # the stand-in for the feature-space search: the nearest target point within the maximum magnitude
            idx, d2 = engine.nn_query(tgt, src, 1, return_d2=True)
            corr = torch.where(torch.sqrt(d2[:, 0]) <= float(self.para.max_magnitude), idx[:, 0].to(torch.int64),
                               torch.full((src.shape[0],), -1, dtype=torch.int64, device=src.device))
        self.data_interim.corres_3d_voxel_from_3d_idx = torch.stack([torch.arange(src.shape[0], device=src.device), corr], dim=1)
        m2 = _get(self.config, "point_matches_from_2d")
        self.data_interim.corres_3d_from_2d_idx = None if m2 is None else torch.stack(
            [torch.arange(src.shape[0], device=src.device), torch.as_tensor(m2, dtype=torch.int64, device=src.device)], dim=1)
        if self.matching != "only_3d" and self.data_interim.corres_3d_from_2d_idx is None:
            raise NotImplementedError(
                f"fine_matching_{self.matching} needs the 3D matches lifted from the 2D image matching (`corres_3d_from_2d_idx`, "
                "base:1670-1675): pass them as cfg.point_matches_from_2d, or set method.fine_matching_only_3d")

    # ---- base:2925-3233 (stand-in / hook) --------------------------------------------------------------------------------------
    def coarse_matching_with_different_types(self):
        """Patch matches as CSR over `spt_corres_src[i]` / `spt_corres_tgt[i]`: cfg.patch_matches_fn(self) -> (src_patch (M,),
        tgt_patch (M,)) label pairs, else every source patch with the target patch most of its 3D point matches lead to."""
        import torch
        I = self.data_interim
        fn = _get(self.config, "patch_matches_fn")
        lab_s, lab_t = I.idx_pts2spt_src.long(), I.idx_pts2spt_tgt.long()
        Ks, Kt = I.spt_off_src.shape[0] - 1, I.spt_off_tgt.shape[0] - 1
        if fn is not None:
            ps, pt = (torch.as_tensor(v, dtype=torch.int64, device=lab_s.device) for v in fn(self))
        else:
            corr = I.corres_3d_voxel_from_3d_idx[:, 1]
            has = corr >= 0
            pair = lab_s[has] * Kt + lab_t[corr[has]]
            uniq, cnt = torch.unique(pair, return_counts=True)
            ps_all = uniq // Kt
            # the most frequent target patch of every source patch (ties: the smallest label)
            best = torch.zeros(Ks, dtype=torch.int64, device=lab_s.device)
            best.scatter_reduce_(0, ps_all, cnt, "amax", include_self=True)
            win = cnt == best[ps_all]
            first = torch.full((Ks,), Kt, dtype=torch.int64, device=lab_s.device)
            first.scatter_reduce_(0, ps_all[win], (uniq % Kt)[win], "amin", include_self=True)
            ps = torch.nonzero(first < Kt, as_tuple=True)[0]
            pt = first[ps]
        minimum = int(_get(self.method, "num_min_matches_for_small_patch", 0)) if _get(self.method, "small_patch_removal", False) else 0
        n_s = (I.spt_off_src[1:] - I.spt_off_src[:-1])[ps]
        n_t = (I.spt_off_tgt[1:] - I.spt_off_tgt[:-1])[pt]
        keep = (n_s >= minimum) & (n_t >= minimum)
        ps, pt, n_s, n_t = ps[keep], pt[keep], n_s[keep], n_t[keep]

        def gather(order, off, patches, counts):
            out_off = torch.zeros(patches.shape[0] + 1, dtype=torch.int64, device=order.device)
            out_off[1:] = torch.cumsum(counts, 0)
            pid = torch.repeat_interleave(torch.arange(patches.shape[0], device=order.device), counts)
            pos = torch.arange(int(out_off[-1]), device=order.device) - out_off[pid] + off[patches][pid]
            return order[pos], out_off

        O = self.data_output
        O.spt_corres_src_ids, O.spt_corres_src_off = gather(I.spt_order_src, I.spt_off_src, ps, n_s)
        # (ids ascending inside a target patch: f4l_labels_to_csr sorts stably by label)
        O.spt_corres_tgt_ids, O.spt_corres_tgt_off = gather(I.spt_order_tgt, I.spt_off_tgt, pt, n_t)
        O.spt_match_src_label, O.spt_match_tgt_label = ps, pt
        if self.verbose:
            self.logging.info(f"Coarse matching is done! {ps.shape[0]} patch matches of {Ks} source / {Kt} target patches")

    # ---- base:3236-3436 --------------------------------------------------------------------------------------------------------
    def fine_matching_with_different_types(self):
        self.fine_matching_prepare()
        a = self.fine_state["loop_args"]
        out = engine.patch_loop(a["src"], a["src_off"], a["tgt"], a["tgt_off"], a["corr_src"], a["corr_ref"], a["corr_off"], a["corr_weights"],
                                0.0, 1e-6, rows_src=a["rows_src"], rows_off=a["rows_off"], **self.fine_state["loop_kw"])
        self.fine_matching_finish(out)

    def fine_matching_prepare(self):
        """Everything of base:3236-3436 before the per-patch loop; leaves `self.fine_state` (fine_matching.fine_matching_prepare), whose
        `loop_args` / `loop_kw` one f4l_patch_loop launch takes -- this tile's alone, or several tiles' merged (main_fusion)."""
        I, O, m = self.data_interim, self.data_output, self.method
        if not _get(m, "icp_refine", True):
            raise NotImplementedError("icp_refine: False leaves the reference without an output (base:3441-3442)")
        c2d = None if I.corres_3d_from_2d_idx is None else I.corres_3d_from_2d_idx[:, 1]
        self.fine_state = fine_matching_prepare(
            self.data_input_3d.src_pts, self.data_input_3d.tgt_pts, O.spt_corres_src_ids, O.spt_corres_src_off, O.spt_corres_tgt_ids,
            O.spt_corres_tgt_off, I.corres_3d_voxel_from_3d_idx[:, 1], corr_tgt_2d=c2d, matching=self.matching,
            weighting_svd=bool(_get(m, "weighting_svd", False)), num_min_fine_match=int(m.num_min_fine_match),
            icp_threshold=float(self.para.icp_threshold), remove_low_quality_patch_matches=bool(_get(m, "remove_low_quality_patch_matches", False)),
            num_min_matches_for_quality_check=int(_get(m, "num_min_matches_for_quality_check", 10)), thres_dist_diff=float(_get(m, "thres_dist_diff", 0.1)),
            thres_inlier_ratio=float(_get(m, "thres_inlier_ratio", 0.5)), assign_type=m.assign_type, output_tgt2src=bool(_get(m, "output_tgt2src", False)),
            median_max_resolution=float(self.para.median_max_resolution),
            rigidity_precision=str(_get(m, "rigidity_precision", "f64")))  # (this build's key; the reference has no such choice)

    def fine_matching_finish(self, out):
        O = self.data_output
        res = fine_matching_finish(self.fine_state, out)
        self.fine_state = None
        O.fine = res
        O.corres_3d_refine_apply_icp = res["dense"]
        O.corres_3d_refine_apply_icp_discrete = res["sparse"]
        O.corres_3d_refine_apply_icp_tgt2src = res["tgt2src"]
        if self.verbose:
            self.logging.info(f"Fine matching is done! {int((res['iters'] >= 0).sum())} of {res['iters'].shape[0]} patch matches registered")

    def save_process_dvf(self):
        O = self.data_output
        return save_process_dvf(self.output_root, self.tile_id, self.data.dataset, O.corres_3d_refine_apply_icp,
                                O.corres_3d_refine_apply_icp_discrete, O.corres_3d_refine_apply_icp_tgt2src,
                                multiple_case=bool(self.data.multiple_case), voxel_size=_get(self.method, "voxel_size"), defer=self.defer_files)

    # ---- src/coarse_to_fine_matching.py:201-290 -------------------------------------------------------------------------------
    def implement_c2f_matching(self):
        self.prepare_c2f()
        a = self.fine_state["loop_args"]
        return self.finish_c2f(engine.patch_loop(a["src"], a["src_off"], a["tgt"], a["tgt_off"], a["corr_src"], a["corr_ref"], a["corr_off"],
                                                 a["corr_weights"], 0.0, 1e-6, rows_src=a["rows_src"], rows_off=a["rows_off"],
                                                 **self.fine_state["loop_kw"]))

    # The same in two halves around the per-patch loop, so that the loop of SEVERAL tiles can be one launch (a <= 1 M-point tile's
    # ~2000 patch matches are two rounds of workgroups on an MI355X: main_fusion's tile loop merges them, utils/tiles.py).
    def prepare_c2f(self):
        if self.verbose:
            self.logging.info("Skip 2d matching!" if self.matching == "only_3d" else
                              "2d matching is not run here: its lifted matches come in as cfg.point_matches_from_2d")
        self._voxel_subsampling()
        self.implement_partition()
        self.load_partition()
        self.prepare_pts2spt_dict()
        self.global_matches_from_3d()
        self.coarse_matching_with_different_types()
        self.fine_matching_prepare()
        return self

    def finish_c2f(self, out):
        self.fine_matching_finish(out)
        if self.data_output.corres_3d_refine_apply_icp.shape[0]:
            self.written = self.save_process_dvf()
        return self.data_output
