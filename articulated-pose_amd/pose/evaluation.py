"""The last two steps of the reference's evaluation.sh, batched on the GPU:

    evaluation/eval_pose_err.py    error tables (:111-172), amodal-box boundaries (:210-277), relative joint-state errors (:279-363)
    evaluation/compute_miou.py     per-part 3-D IoU of the amodal boxes (:150-240)

The reference walks the records one frame, one key ('baseline' / 'nonlinear') and one part at a time in numpy; here the frames of a key
are stacked and every per-point reduction (`ancsh_part_extents`: predicted labels, NOCS extents, dynamic boundary) and every 50^3-grid IoU
(`ancsh_iou_3d`) is one launch over all frames.  The skip rules are the scripts' own: a failed fit (scale None), a NaN translation, a
record missing from the ground-truth pickles, a part nobody is predicted to belong to (the reference's np.max raises inside a bare
`except: pass`) drop the frame for that key.
`datas` = {'pn_gt', 'gn_gt', 'baseline', 'nonlinear'} -> {basename: record}; `load(exp, basename)` -> dict of arrays of
results/test_pred/<exp>/<basename>.{h5,npz}."""
import numpy as np
import torch

from .. import _lib
from . import metrics as M

KEYS = ("baseline", "nonlinear")


def compose_rt(rotation, translation):
    """eval_pose_err.py:25-30 (float32 4 x 4)"""
    m = np.zeros((4, 4), dtype=np.float32)
    m[:3, :3] = np.asarray(rotation)[:3, :3]
    m[:3, 3] = translation
    m[3, 3] = 1
    return m


def part_extents(nocs, mask, P, r0, t0):
    """nocs (B,N,3K|3), mask (B,N,K), P (B,N,>=3) float32 CUDA tensors; r0 (B,3,3), t0 (B,3): part 0's fitted pose.
    -> scale_pred (B,K,3) float32, dynam (B,K) float64, count (B,K) int32 (ancsh_part_extents)."""
    nocs, mask, P = nocs.contiguous().float(), mask.contiguous().float(), P.contiguous().float()
    _lib.require_cuda(nocs, mask, P)
    B, N, K = mask.shape
    if nocs.shape[:2] != (B, N) or P.shape[:2] != (B, N) or nocs.shape[2] not in (3, 3 * K) or P.shape[2] < 3:
        raise ValueError("part_extents: nocs %s / mask %s / P %s do not go together" % (tuple(nocs.shape), tuple(mask.shape), tuple(P.shape)))
    dev = nocs.device
    pose0 = torch.cat([torch.as_tensor(r0, dtype=torch.float64, device=dev).reshape(B, 9),
                       torch.as_tensor(t0, dtype=torch.float64, device=dev).reshape(B, 3)], dim=1).contiguous()
    scale = torch.empty((B, K, 3), dtype=torch.float32, device=dev)
    dynam = torch.empty((B, K), dtype=torch.float64, device=dev)
    count = torch.empty((B, K), dtype=torch.int32, device=dev)
    _lib.call("ancsh_part_extents", B, N, K, nocs.shape[2], _lib.ptr(nocs), _lib.ptr(mask), _lib.ptr(P), P.shape[2], _lib.ptr(pose0),
              _lib.ptr(scale), _lib.ptr(dynam), _lib.ptr(count))
    return scale, dynam, count


def raw_errors(datas, skip_instances=()):
    """eval_pose_err.py:111-126 / compute_miou.py:100-113: the rows the fit itself wrote (rpy_err, xyz_err), failed fits left out."""
    r, t = {k: [] for k in KEYS}, {k: [] for k in KEYS}
    for key in KEYS:
        for basename, rec in datas[key].items():
            if basename.split('_')[0] in skip_instances or rec['scale'] is None or rec['scale'] is []:
                continue
            r[key].append(rec['rpy_err'][key])
            t[key].append(rec['xyz_err'][key])
    return r, t


def _table(lines, title, domain, nocs, rows):
    lines.append('For {} object, {} nocs, {} per part is: '.format(domain, nocs, title))
    for k in KEYS:
        lines.append(k[0:8] + ' ' + ' '.join('{:0.4f}'.format(float(x)) for x in rows[k]))
    lines.append('\n')


def error_report(r_raw, t_raw, num_parts, domain, nocs, device="cuda:0"):
    """eval_pose_err.py:128-172 -> the printed lines of the four tables (NaN translation errors count as 0)."""
    lines, rows = [], {t: {} for t in range(4)}
    for k in KEYS:
        r = torch.as_tensor(np.array(r_raw[k], dtype=np.float64).reshape(-1, num_parts), device=device)
        t = torch.nan_to_num(torch.as_tensor(np.array(t_raw[k], dtype=np.float64).reshape(-1, num_parts), device=device), nan=0.0)
        nv = max(r.shape[0], 1)
        ok = r < 5
        rows[0][k] = (r.sum(0) / nv).cpu().numpy()
        rows[1][k] = (t.sum(0) / nv).cpu().numpy()
        rows[2][k] = (ok.sum(0).double() / nv).cpu().numpy()
        rows[3][k] = ((ok & (t < 0.05)).sum(0).double() / nv).cpu().numpy()
    for i, title in enumerate(('mean rotation err', 'mean translation err', '5 degrees accuracy', '5 degrees, 5 cms accuracy')):
        _table(lines, title, domain, nocs, rows[i])
    return lines


def urdf_joint_rpy(path):
    """The one thing the evaluation reads from a SAPIEN mobility.urdf (lib/data_utils.py:230-321, get_urdf_mobility): rpy of the origin of
    joint_<k>, k = 0 .. links - 2 ([0, 0, 0] when the origin has none).  path: the instance's directory."""
    import os
    import xml.etree.ElementTree as ET
    root = ET.parse(os.path.join(path, 'mobility.urdf')).getroot()
    rpy = [None] * (len(root.findall('link')) - 1)
    for joint in root.iter('joint'):
        k = int(joint.attrib['name'].split('_')[1])
        for origin in joint.iter('origin'):
            rpy[k] = [float(x) for x in origin.attrib['rpy'].split()] if 'rpy' in origin.attrib else [0, 0, 0]
    return rpy


def euler_matrix_sxyz(ai, aj, ak):
    """Rz(ak) Ry(aj) Rx(ai): lib/transformations.py:1049-1108 with its default axes='sxyz', 3 x 3."""
    import math
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    return np.array([[cj * ck, sj * sc - cs, sj * cc + ss], [cj * sk, sj * ss + cc, sj * cs - sc], [-sj, cj * si, cj * ci]])


def gt_boxes(factors, corners, instances, num_parts, joint_frames=None, spec_map=None):
    """eval_pose_err.py:175-204 / compute_miou.py:116-142: NOCS box corners of every part of every instance from the dataset's
    normalisation tables.  The prismatic 'drawer' passes joint_frames = {instance: rpy list of its URDF joints} and global_info's
    spec_map: both corners are rotated about the box centre by the frame of the joint of part 0 (:184-189, :199-201) and part p is
    stored at position spec_map[instance].index(p)."""
    out = {}
    for ins in instances:
        per_part = [None] * num_parts
        if joint_frames is not None:
            order = spec_map[ins]
            rot_mat = euler_matrix_sxyz(*joint_frames[ins][order[0]])
        for p in range(num_parts):
            nf, nc = factors[ins][p + 1], np.asarray(corners[ins][p + 1])
            c = np.copy(nc)
            c[0] = np.array([0.5, 0.5, 0.5]).reshape(1, 3) - 0.5 * (nc[1] - nc[0]) * nf
            c[1] = np.array([0.5, 0.5, 0.5]).reshape(1, 3) + 0.5 * (nc[1] - nc[0]) * nf
            if joint_frames is not None:
                c[0] = np.dot(c[0].reshape(1, 3) - 0.5, rot_mat.T) + 0.5
                c[1] = np.dot(c[1].reshape(1, 3) - 0.5, rot_mat.T) + 0.5
                per_part[order.index(p)] = c
            else:
                per_part[p] = c
        out[ins] = per_part
    return out


def _usable(cur, basename, key):
    rec = cur.get(basename)
    return not (rec is None or rec['scale'] is None or rec['scale'] is [] or np.any(np.isnan(np.asarray(rec['translation'][key], dtype=np.float64))))


def _frames(datas, key, load, exp_of_key, nocs_field, num_parts, need_gt, device):
    """Stack the usable frames of `key` (order of datas['nonlinear']) and run the per-point reductions in one launch per cloud size.
    -> dict(names, scale_pred (F,K,3) f32, dynam (F,K) f64, r (F,K,3,3), t (F,K,3), s (F,K)) with the frames of empty parts removed."""
    names = [b for b in datas['nonlinear'] if b in datas[key] and _usable(datas[key], b, key)
             and (not need_gt or (b in datas['pn_gt'] and b in datas['gn_gt']))]
    out = dict(names=[], scale_pred=[], dynam=[], r=[], t=[], s=[])
    by_n = {}
    recs = {}
    for b in names:
        # the scripts read a frame's record inside the frame's bare try / except (eval_pose_err.py:215-275, compute_miou.py:152-229): a
        # record that is missing from <exp> or lacks a field drops THAT frame (and, after a baseline failure, its nonlinear pass:
        # _drop_after_baseline_failure), it does not abort the run
        try:
            f = load(exp_of_key, b)
            recs[b] = dict(nocs=np.asarray(f[nocs_field], np.float32), mask=np.asarray(f['instance_per_point'], np.float32),
                           P=np.asarray(f['P'], np.float32))
            if recs[b]['nocs'].shape[0] != recs[b]['P'].shape[0] or recs[b]['mask'].shape != (recs[b]['P'].shape[0], num_parts):
                raise ValueError("record %s: inconsistent shapes" % b)
        except (OSError, KeyError, ValueError):
            recs.pop(b, None)
            continue
        by_n.setdefault(recs[b]['P'].shape[0], []).append(b)
    keep = {}
    for n, group in by_n.items():
        nocs = torch.as_tensor(np.stack([recs[b]['nocs'] for b in group]), device=device)
        mask = torch.as_tensor(np.stack([recs[b]['mask'] for b in group]), device=device)
        P = torch.as_tensor(np.stack([recs[b]['P'] for b in group]), device=device)
        r0 = np.stack([np.asarray(datas[key][b]['rotation'][key][0], np.float64) for b in group])
        t0 = np.stack([np.asarray(datas[key][b]['translation'][key][0], np.float64).reshape(3) for b in group])
        sc, dy, cnt = part_extents(nocs, mask, P, r0, t0)
        sc, dy, cnt = sc.cpu().numpy(), dy.cpu().numpy(), cnt.cpu().numpy()
        for i, b in enumerate(group):
            if (cnt[i] > 0).all():
                keep[b] = (sc[i], dy[i])
    for b in names:
        if b not in keep:
            continue
        rec = datas[key][b]
        out['names'].append(b)
        out['scale_pred'].append(keep[b][0])
        out['dynam'].append(keep[b][1])
        out['r'].append(np.stack([np.asarray(x, np.float64) for x in rec['rotation'][key]]))
        out['t'].append(np.stack([np.asarray(x, np.float64).reshape(3) for x in rec['translation'][key]]))
        out['s'].append(np.array([float(np.asarray(x).reshape(-1)[0]) for x in rec['scale'][key]], np.float64))
    for k in ('scale_pred', 'dynam', 'r', 't', 's'):
        out[k] = np.stack(out[k]) if out[k] else np.zeros((0, num_parts) + {'scale_pred': (3,), 'r': (3, 3), 't': (3,)}.get(k, ()))
    return out


def _drop_after_baseline_failure(datas, per_key, rows=None, missing_raises=False):
    """Both scripts run 'baseline' then 'nonlinear' of a frame inside ONE try: a frame whose baseline pass was attempted (usable record) and
    raised -- missing ground truth, a part without predicted points -- never reaches its nonlinear pass.  missing_raises: compute_miou.py
    indexes the baseline pickle without asking first (:157), so a frame the baseline pickle does not hold raises too."""
    for b in list(per_key['nonlinear']):
        if (_usable(datas['baseline'], b, 'baseline') and b not in per_key['baseline']) or (missing_raises and b not in datas['baseline']):
            if rows is not None:
                rows['nonlinear'].pop(list(per_key['nonlinear']).index(b))
            del per_key['nonlinear'][b]


def boundaries(datas, load, exp, baseline_exp, num_parts, device="cuda:0"):
    """eval_pose_err.py:210-277: boundary_all[key][basename] = {'canon': [...], 'dynam': [...]}.  The baseline reads the mixed network's
    global NOCS (gocs_per_point of <exp>), this path's records the part NOCS of <baseline_exp>."""
    out = {k: {} for k in KEYS}
    for key in KEYS:
        fr = _frames(datas, key, load, baseline_exp if key == 'nonlinear' else exp, 'nocs_per_point' if key == 'nonlinear' else 'gocs_per_point',
                     num_parts, True, device)
        canon = -fr['scale_pred'][:, :, 0] / np.float32(2) + np.float32(0.5)        # float32, like - scale_pred[0] / 2 + 0.5
        for i, b in enumerate(fr['names']):
            out[key][b] = {'canon': [canon[i, j] for j in range(num_parts)], 'dynam': [fr['dynam'][i, j] for j in range(num_parts)]}
    _drop_after_baseline_failure(datas, out)
    return out


def relative_errors(datas, boundary_all, num_parts, nocs='ANCSH', device="cuda:0"):
    """eval_pose_err.py:279-338: per record and joint j = 1..K-1 the error of the relative rotation R0^T Rj (degrees) and of the relative
    translation -- the boundary difference along part 0's x axis -- against the ground truth's."""
    r_out, t_out = {k: [] for k in KEYS}, {k: [] for k in KEYS}
    for key in KEYS:
        cur = datas[key]
        # (eval_pose_err.py:296-312: with --nocs NAOCS the nonlinear pass takes t1 - t0 and never reads boundary_all, so a frame without
        # a boundary entry still counts there; every other pass indexes boundary_all inside its try)
        uses_boundary = not (nocs == 'NAOCS' and key == 'nonlinear')
        names = [b for b in datas['nonlinear'] if b in cur and _usable(cur, b, key) and (b in boundary_all[key] or not uses_boundary)
                 and datas['pn_gt'][b]['rt'] is not None and datas['pn_gt'][b]['scale'] is not None]
        if not names:
            continue
        f64 = dict(dtype=torch.float64, device=device)
        r = torch.as_tensor(np.stack([np.stack([np.asarray(x, np.float64) for x in cur[b]['rotation'][key]]) for b in names]), **f64)
        t = torch.as_tensor(np.stack([np.stack([np.asarray(x, np.float64).reshape(3) for x in cur[b]['translation'][key]]) for b in names]), **f64)
        rt_p = torch.as_tensor(np.stack([np.stack(datas['pn_gt'][b]['rt']['gt']) for b in names]).astype(np.float32), device=device)
        rt_g = torch.as_tensor(np.stack([np.stack(datas['gn_gt'][b]['rt']['gt']) for b in names]).astype(np.float32), device=device)
        d = torch.as_tensor(np.array([[float(boundary_all[key][b]['dynam'][j]) - float(boundary_all[key][b]['canon'][j]) if uses_boundary else 0.0
                                       for j in range(num_parts)] for b in names]), **f64)
        r_diff_pred = r[:, :1].transpose(-1, -2) @ r[:, 1:]                                     # (F, K-1, 3, 3)
        r_diff_gt = (rt_p[:, :1, :3, :3].transpose(-1, -2) @ rt_p[:, 1:, :3, :3]).double()      # float32 matmul, like the reference's
        r_err = M.rot_diff_degree_batch(r_diff_gt, r_diff_pred)
        if nocs == 'NAOCS' and key == 'nonlinear':
            t_diff_pred = t[:, 1:] - t[:, :1]
        else:
            t_diff_pred = d[:, 1:, None] * r[:, :1, :, 0]                                       # R0 . [d, 0, 0]
        t_diff_gt = (rt_g[:, 1:, :3, 3] - rt_g[:, :1, :3, 3]).double()                          # float32 subtraction
        t_err = (t_diff_gt - t_diff_pred).norm(dim=-1)
        r_out[key] = r_err.cpu().numpy().tolist()
        t_out[key] = t_err.cpu().numpy().tolist()
    return r_out, t_out


def relative_report(r_diff, t_diff, num_parts, item, domain, nocs):
    """eval_pose_err.py:340-363"""
    lines, rows = [], {}
    which, title = (t_diff, 'mean relative translation err') if item == 'drawer' else (r_diff, 'mean relative rotation err')
    for k in KEYS:
        a = np.nan_to_num(np.array(which[k], dtype=np.float64).reshape(-1, num_parts - 1), nan=0.0) if item != 'drawer' else \
            np.array(which[k], dtype=np.float64).reshape(-1, num_parts - 1)
        rows[k] = a.sum(0) / max(a.shape[0], 1)
    _table(lines, title, domain, nocs, rows)
    return lines


def miou(datas, load, baseline_exp, bbox3d_all, num_parts, device="cuda:0"):
    """compute_miou.py:150-229: iou_rat[key] = rows of per-part 3-D IoU between the ground-truth box posed by the ground-truth pose and the
    predicted NOCS extents posed by the fitted (s, R, t); both keys read the part-NOCS network's record.  One ancsh_iou_3d launch per key."""
    iou_rat, boundary_all = {k: [] for k in KEYS}, {k: {} for k in KEYS}
    for key in KEYS:
        fr = _frames(datas, key, load, baseline_exp, 'nocs_per_point', num_parts, True, device)
        F = len(fr['names'])
        if F == 0:
            continue
        f64 = dict(dtype=torch.float64, device=device)
        scale_gt = np.stack([[np.asarray(bbox3d_all[b.split('_')[0]][j][1][0]) - np.asarray(bbox3d_all[b.split('_')[0]][j][0][0])
                              for j in range(num_parts)] for b in fr['names']])
        s_gt = np.array([[float(np.asarray(datas['pn_gt'][b]['scale']['gt'][j]).reshape(-1)[0]) for j in range(num_parts)] for b in fr['names']])
        rt_gt = torch.as_tensor(np.stack([np.stack(datas['pn_gt'][b]['rt']['gt']) for b in fr['names']]).astype(np.float32), device=device).double()
        box_gt = M.amodal_boxes(torch.as_tensor(scale_gt, **f64), torch.as_tensor(s_gt, **f64), rt_gt[..., :3, :3], rt_gt[..., :3, 3])
        # the predicted pose goes through compose_rt: float32
        r32 = torch.as_tensor(fr['r'], **f64).float().double()
        t32 = torch.as_tensor(fr['t'], **f64).float().double()
        box_pred = M.amodal_boxes(torch.as_tensor(fr['scale_pred'], device=device).double(), torch.as_tensor(fr['s'], **f64), r32, t32)
        iou = M.iou_3d_batch(box_gt.reshape(F * num_parts, 8, 3), box_pred.reshape(F * num_parts, 8, 3)).reshape(F, num_parts).cpu().numpy()
        iou_rat[key] = iou.tolist()
        canon = -fr['scale_pred'][:, :, 0] / np.float32(2) + np.float32(0.5)
        for i, b in enumerate(fr['names']):
            boundary_all[key][b] = {'canon': [canon[i, j] for j in range(num_parts)], 'dynam': [fr['dynam'][i, j] for j in range(num_parts)]}
    _drop_after_baseline_failure(datas, boundary_all, iou_rat, missing_raises=True)
    return iou_rat, boundary_all


def miou_report(iou_rat, num_parts, domain, nocs):
    """compute_miou.py:231-240"""
    lines, rows = [], {}
    for k in KEYS:
        a = np.array(iou_rat[k], dtype=np.float64).reshape(-1, num_parts)
        rows[k] = a.sum(0) / max(a.shape[0], 1)
    _table(lines, '3D IoU', domain, nocs, rows)
    return lines
