"""CPU ORACLE of the last two steps of the reference's evaluation.sh (test infrastructure only; never imported by the product).

numpy restatement, frame by frame and part by part like the scripts themselves, of
    evaluation/eval_pose_err.py:91-364     error tables, accuracies, amodal-box boundaries, relative (joint-state) errors
    evaluation/compute_miou.py:76-241      per-part 3-D IoU of the amodal boxes
    evaluation/eval_joint_params.py:78-270 per-joint axis-angle / line-distance errors (script level; the per-sample body is joint_params_oracle.py)
PINNED: tests/golden/gen_eval_scripts_golden.py RUNS the scripts (runpy, unmodified, where they lie) on a synthetic results tree in the
reference's directory layout and stores their printed reports and final variables in tests/golden/eval_scripts.pkl;
tests/test_eval_scripts_cpu.py requires the functions below to reproduce those variables exactly and the reports character by character.

`datas` = {'pn_gt', 'gn_gt', 'baseline', 'nonlinear'} -> {basename: record} as the scripts assemble it (:91-109 / :76-99);
`load(exp, basename)` returns the prediction record of results/test_pred/<exp>/<basename>.h5 as a dict of arrays."""
import numpy as np

from oracle import metrics_oracle as MO

KEYS = ("baseline", "nonlinear")


def compose_rt(rotation, translation):
    """eval_pose_err.py:25-30"""
    m = np.zeros((4, 4), dtype=np.float32)
    m[:3, :3] = rotation[:3, :3]
    m[:3, 3] = translation
    m[3, 3] = 1
    return m


def raw_errors(datas, skip_instances=()):
    """eval_pose_err.py:111-126 (skips instance '45841') / compute_miou.py:100-113 (skips nothing): rows = records whose fit succeeded."""
    r, t = {k: [] for k in KEYS}, {k: [] for k in KEYS}
    for key in KEYS:
        for basename, rec in datas[key].items():
            if basename.split('_')[0] in skip_instances:
                continue
            if rec['scale'] is None or rec['scale'] is []:
                continue
            r[key].append(rec['rpy_err'][key])
            t[key].append(rec['xyz_err'][key])
    return r, t


def error_report(r_raw, t_raw, num_parts, domain, nocs):
    """eval_pose_err.py:128-172: the four printed tables, as the list of printed lines (NaN translation errors count as 0)."""
    lines = []
    r = {k: np.array(r_raw[k]) for k in KEYS}
    t = {k: np.array(t_raw[k]) for k in KEYS}
    for k in KEYS:
        t[k][np.where(np.isnan(t[k]))] = 0

    def table(title, fn):
        lines.append('For {} object, {} nocs, {} per part is: '.format(domain, nocs, title))
        for k in KEYS:
            nv = r[k].shape[0]
            lines.append(k[0:8] + ' ' + ' '.join('{:0.4f}'.format(fn(k, j, nv)) for j in range(num_parts)))
        lines.append('\n')

    table('mean rotation err', lambda k, j, nv: np.sum(r[k][:, j]) / nv)
    table('mean translation err', lambda k, j, nv: np.sum(t[k][:, j]) / nv)
    table('5 degrees accuracy', lambda k, j, nv: len(np.where(r[k][:, j] < 5)[0]) / nv)
    table('5 degrees, 5 cms accuracy', lambda k, j, nv: len(np.where(t[k][np.where(r[k][:, j] < 5)[0], j] < 0.05)[0]) / nv)
    return lines


def urdf_joint_rpy(text):
    """lib/data_utils.py:230-321 (get_urdf_mobility) reduced to what the evaluation reads: rpy of joint_<k>'s origin, k = 0 .. links - 2."""
    import xml.etree.ElementTree as ET
    root = ET.fromstring(text)
    rpy = [None] * (len(root.findall('link')) - 1)
    for joint in root.iter('joint'):
        k = int(joint.attrib['name'].split('_')[1])
        for origin in joint.iter('origin'):
            rpy[k] = [float(x) for x in origin.attrib['rpy'].split()] if 'rpy' in origin.attrib else [0, 0, 0]
    return rpy


def euler_matrix_sxyz(ai, aj, ak):
    """lib/transformations.py:1049-1108 for axes='sxyz' (static x, y, z: Rz(ak) Ry(aj) Rx(ai)), 3 x 3 part."""
    import math
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    cc, cs = ci * ck, ci * sk
    sc, ss = si * ck, si * sk
    M = np.identity(4)        # a 3 x 3 VIEW of a 4 x 4 matrix, like the reference's euler_matrix(...)[:3, :3]: np.dot takes another path for it
    M[:3, :3] = [[cj * ck, sj * sc - cs, sj * cc + ss], [cj * sk, sj * ss + cc, sj * cs - sc], [-sj, cj * si, cj * ci]]
    return M[:3, :3]


def gt_boxes(factors, corners, instances, num_parts, urdf=None, spec_map=None):
    """eval_pose_err.py:175-204 / compute_miou.py:116-142: the NOCS box corners of every part of every instance.  The prismatic 'drawer'
    (urdf = {instance: mobility.urdf text}, spec_map = global_info's part order) rotates both corners about the box centre by the URDF
    frame of the joint of its part 0 and stores part p at position target_order.index(p)."""
    out = {}
    for ins in instances:
        per_part = [None] * num_parts
        if urdf is not None:
            order = spec_map[ins]
            rot_mat = euler_matrix_sxyz(*urdf_joint_rpy(urdf[ins])[order[0]])
        for p in range(num_parts):
            nf, nc = factors[ins][p + 1], corners[ins][p + 1]
            c = np.copy(nc)
            c[0] = np.array([0.5, 0.5, 0.5]).reshape(1, 3) - 0.5 * (nc[1] - nc[0]) * nf
            c[1] = np.array([0.5, 0.5, 0.5]).reshape(1, 3) + 0.5 * (nc[1] - nc[0]) * nf
            if urdf is not None:
                c[0] = np.dot(c[0].reshape(1, 3) - 0.5, rot_mat.T) + 0.5
                c[1] = np.dot(c[1].reshape(1, 3) - 0.5, rot_mat.T) + 0.5
                per_part[order.index(p)] = c
            else:
                per_part[p] = c
        out[ins] = per_part
    return out


def frame_parts(rec, r, t, s, s_gt, bbox_gt, num_parts):
    """The per-part block both scripts share (eval_pose_err.py:253-274, compute_miou.py:196-211): predicted labels, NOCS extents of the
    predicted part, the canonical boundary, the dynamic boundary (minimum x of the part's points taken back through part 0's pose), the
    ground-truth box, scale / volume errors."""
    nocs_pred, pts = rec['nocs'], rec['P']
    cls_pred = np.argmax(rec['instance_per_point'], axis=1)
    rt_0 = compose_rt(r[0], t[0])
    out = dict(scale_pred=[], canon=[], dynam=[], box_pred=[], box_gt=[], scale_err=[], volume_err=[])
    for j in range(num_parts):
        idx = np.where(cls_pred == j)[0]
        if nocs_pred.shape[1] == 3:
            centered = nocs_pred[idx, :3] - 0.5
        else:
            centered = nocs_pred[idx, 3 * j:3 * (j + 1)] - 0.5
        scale_pred = 2 * np.max(abs(centered), axis=0)
        out['scale_pred'].append(scale_pred)
        out['box_pred'].append(MO.get_3d_bbox(scale_pred, shift=np.array([1 / 2, 1 / 2, 1 / 2])).transpose())
        out['canon'].append(- scale_pred[0] / 2 + 0.5)
        shifted = np.dot(np.concatenate([pts[idx, :3], np.ones((len(idx), 1))], axis=1), np.linalg.pinv(rt_0.T))
        out['dynam'].append(np.min(shifted[:, 0]))
        scale_gt = bbox_gt[j][1][0] - bbox_gt[j][0][0]
        out['box_gt'].append(MO.get_3d_bbox(scale_gt, shift=np.array([1 / 2, 1 / 2, 1 / 2])).transpose())
        out['scale_err'].append(np.linalg.norm(scale_pred * s[j] - scale_gt * s_gt[j]))
        out['volume_err'].append(scale_pred[0] * scale_pred[1] * scale_pred[2] * s[j] / (scale_gt[0] * scale_gt[1] * scale_gt[2] * s_gt[j][0]) - 1)
    return out


def _usable(cur, basename, key):
    rec = cur.get(basename)
    return not (rec is None or rec['scale'] is None or rec['scale'] is [] or np.any(np.isnan(rec['translation'][key])))


def boundaries(datas, load, exp, baseline_exp, bbox3d_all, num_parts):
    """eval_pose_err.py:210-277: boundary_all[key][basename] = {'canon': [...], 'dynam': [...]} over the records of 'nonlinear'.
    The baseline reads the mixed network's global NOCS (gocs_per_point of <exp>), ours the part NOCS of <baseline_exp>."""
    out = {k: {} for k in KEYS}
    for basename in datas['nonlinear']:
        try:                                                  # the script's bare try / except: pass around BOTH keys of a record
            for key in KEYS:
                cur = datas[key]
                if not _usable(cur, basename, key):
                    continue
                rt_g = datas['gn_gt'][basename]['rt']['gt']   # (:232-236: KeyError when the record has no ground truth)
                f = load(baseline_exp if key == 'nonlinear' else exp, basename)
                rec = dict(nocs=f['nocs_per_point'] if key == 'nonlinear' else f['gocs_per_point'], P=f['P'], instance_per_point=f['instance_per_point'])
                r, t, s = cur[basename]['rotation'][key], cur[basename]['translation'][key], cur[basename]['scale'][key]
                fp = frame_parts(rec, r, t, s, datas['pn_gt'][basename]['scale']['gt'], bbox3d_all[basename.split('_')[0]], num_parts)
                out[key][basename] = {'canon': fp['canon'], 'dynam': fp['dynam']}
        except Exception:
            pass
    return out


def relative_errors(datas, boundary_all, num_parts, nocs='ANCSH'):
    """eval_pose_err.py:279-338: per record and joint j = 1..K-1 the error of the relative rotation R0^T Rj and of the relative
    translation (boundary difference along part 0's x axis) against the ground truth's."""
    r_out, t_out = {k: [] for k in KEYS}, {k: [] for k in KEYS}
    for key in KEYS:
        cur = datas[key]
        for basename in datas['nonlinear']:
            rec = cur.get(basename)
            if rec is None:
                continue
            if rec['scale'] is None or rec['scale'] is [] or np.any(np.isnan(rec['translation'][key])) or basename not in boundary_all[key]:
                continue
            if datas['pn_gt'][basename]['rt'] is None or datas['pn_gt'][basename]['scale'] is None:
                continue
            rt_p, rt_g = datas['pn_gt'][basename]['rt']['gt'], datas['gn_gt'][basename]['rt']['gt']
            r, t = rec['rotation'][key], rec['translation'][key]
            r_err, t_err = [], []
            for j in range(1, num_parts):
                r_diff_pred = np.matmul(r[0].T, r[j])
                if nocs == 'NAOCS' and key == 'nonlinear':
                    t_diff_pred = (t[j] - t[0]).reshape(-1)
                else:
                    d = boundary_all[key][basename]['dynam'][j] - boundary_all[key][basename]['canon'][j]
                    t_diff_pred = np.dot(r[0], np.array([d, 0, 0]).reshape(3, 1)).reshape(-1)
                r_diff_gt = np.matmul(rt_p[0][:3, :3].T, rt_p[j][:3, :3])
                t_diff_gt = (rt_g[j][:3, 3] - rt_g[0][:3, 3]).reshape(-1)
                t_err.append(np.linalg.norm(t_diff_gt - t_diff_pred))
                r_err.append(MO.rot_diff_degree(r_diff_gt, r_diff_pred))
            r_out[key].append(r_err)
            t_out[key].append(t_err)
    return r_out, t_out


def relative_report(r_diff, t_diff, num_parts, item, domain, nocs):
    """eval_pose_err.py:340-363"""
    lines = []
    if item == 'drawer':
        lines.append('For {} object, {} nocs, mean relative translation err per part is: '.format(domain, nocs))
        for k in KEYS:
            a = np.array(t_diff[k])
            lines.append(k[0:8] + ' ' + ' '.join('{:0.4f}'.format(np.sum(a[:, j]) / a.shape[0]) for j in range(num_parts - 1)))
    else:
        lines.append('For {} object, {} nocs, mean relative rotation err per part is: '.format(domain, nocs))
        for k in KEYS:
            a = np.array(r_diff[k])
            a[np.where(np.isnan(a))] = 0
            lines.append(k[0:8] + ' ' + ' '.join('{:0.4f}'.format(np.sum(a[:, j]) / a.shape[0]) for j in range(num_parts - 1)))
    lines.append('\n')
    return lines


def miou(datas, load, baseline_exp, bbox3d_all, num_parts):
    """compute_miou.py:150-229: iou_rat[key] = rows of per-part 3-D IoU (ground-truth box posed by the ground-truth pose, predicted NOCS
    extents posed by the fitted (s, R, t)); both keys read the part-NOCS network's record (<baseline_exp>)."""
    iou_rat = {k: [] for k in KEYS}
    boundary_all = {k: {} for k in KEYS}
    for basename in datas['nonlinear']:
        try:                                                  # the script's bare try / except: pass around BOTH keys of a record
            for key in KEYS:
                cur = datas[key]
                if cur[basename]['scale'] is None or cur[basename]['scale'] is [] or np.any(np.isnan(cur[basename]['translation'][key])):
                    continue
                rt_gt, s_gt = datas['pn_gt'][basename]['rt']['gt'], datas['pn_gt'][basename]['scale']['gt']
                rt_g = datas['gn_gt'][basename]['rt']['gt']
                r, t, s = cur[basename]['rotation'][key], cur[basename]['translation'][key], cur[basename]['scale'][key]
                f = load(baseline_exp, basename)
                rec = dict(nocs=f['nocs_per_point'], P=f['P'], instance_per_point=f['instance_per_point'])
                fp = frame_parts(rec, r, t, s, s_gt, bbox3d_all[basename.split('_')[0]], num_parts)
                row = []
                for j in range(num_parts):
                    bb1 = fp['box_gt'][j] * s_gt[j][0]
                    bb2 = fp['box_pred'][j] * s[j]
                    rt1, rt2 = rt_gt[j], compose_rt(r[j], t[j])
                    row.append(MO.iou_3d(np.dot(bb1, rt1[:3, :3].T) + rt1[:3, 3], np.dot(bb2, rt2[:3, :3].T) + rt2[:3, 3]))
                iou_rat[key].append(row)
                boundary_all[key][basename] = {'canon': fp['canon'], 'dynam': fp['dynam']}
        except Exception:
            pass
    return iou_rat, boundary_all


def miou_report(iou_rat, num_parts, domain, nocs):
    """compute_miou.py:231-240"""
    lines = ['For {} object, {} nocs, 3D IoU per part is: '.format(domain, nocs)]
    for k in KEYS:
        a = np.array(iou_rat[k])
        lines.append(k[0:8] + ' ' + ' '.join('{:0.4f}'.format(np.sum(a[:, j]) / a.shape[0]) for j in range(num_parts)))
    lines.append('\n')
    return lines


def joint_param_errors(datas, load, exp, num_parts):
    """evaluation/eval_joint_params.py:104-262 at the script level: every record of datas['nonlinear'] inside the script's bare
    try / except: pass, -> (angle_err_all, dist_err_all) rows of K-1 entries.  The per-sample body is oracle/joint_params_oracle.py
    (pinned line by line by tests/golden/joint_params.npz); the camera-space products keep numpy's own promotion of `s[0] * p`
    (a float64 scale from this build's pickles gives a float64 product, the reference solver's float32 scale a float32 one)."""
    from oracle import joint_params_oracle as JO
    angle_all, dist_all = [], []
    for basename in datas['nonlinear']:
        try:
            f = load(exp, basename)
            jc_pred = np.argmax(f['index_per_point'], axis=1)
            sc, tr, p, l = JO.st_and_joints(f['gocs_per_point'], f['nocs_per_point'], f['instance_per_point'], f['heatmap_per_point'],
                                            f['unitvec_per_point'], f['joint_axis_per_point'], jc_pred, num_parts)
            pg, lg = JO.gt_joints(f['nocs_gt_g'], f['heatmap_gt'], f['unitvec_gt'], f['joint_axis_gt'], f['joint_cls_gt'], num_parts)
            rt_gt, s_gt = datas['pn_gt'][basename]['rt']['gt'], datas['pn_gt'][basename]['scale']['gt']
            rt_g, s_g = datas['gn_gt'][basename]['rt']['gt'], datas['gn_gt'][basename]['scale']['gt']
            r, t, s = (datas['nonlinear'][basename][k]['nonlinear'] for k in ('rotation', 'translation', 'scale'))
            angle_err, dist_err = [], []
            for j in range(1, num_parts):
                tp = p[j - 1] * sc[0] + tr[0]
                cp = np.dot(s[0] * tp.reshape(1, 3), r[0].T) + t[0]
                cl = np.dot(l[j - 1].reshape(1, 3), r[0].T)
                gp = np.dot(s_g[0] * pg[j - 1].reshape(1, 3), rt_g[0][:3, :3].T) + rt_g[0][:3, 3]
                gl = np.dot(lg[j - 1].reshape(1, 3), rt_g[0][:3, :3].T)
                angle_err.append(MO.axis_diff_degree(gl, cl))
                dist_err.append(MO.dist_between_3d_lines(gp, gl, cp, cl))
            if len(angle_err) == num_parts - 1:
                angle_all.append(angle_err)
                dist_all.append(dist_err)
        except Exception:
            pass
    return angle_all, dist_all


def joint_param_report(angle_all, dist_all, num_parts):
    """eval_joint_params.py:263-270: NaNs count as 0; per joint the mean absolute angle (degrees) and line distance."""
    r, t = np.array(angle_all), np.array(dist_all)
    r[np.where(np.isnan(r))] = 0
    t[np.where(np.isnan(t))] = 0
    lines = ['{} {} {}'.format(r.shape, t.shape, num_parts)]
    for k in range(num_parts - 1):
        lines.append('joint {} with mean angle error {} degrees, mean dist {}'.format(k, np.mean(np.abs(r[:, k])), np.mean(np.abs(t[:, k]))))
        lines.append('{} {}'.format(np.mean(np.abs(r[:, k])), np.mean(np.abs(t[:, k]))))
    return lines
