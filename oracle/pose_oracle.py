"""CPU ORACLE of the pose-fit half (test infrastructure only; never imported by the product).

A numpy/scipy restatement of the reference's per-part RANSAC + Kabsch/scale fit + articulated
Levenberg-Marquardt refinement, function by function:

    lib/d3_utils.py:150-163          rotate_points_with_rotvec
    lib/d3_utils.py:206-220          rotate_pts            (Kabsch via 3x3 SVD, reflection fix)
    lib/d3_utils.py:223-234          transform_pts
    lib/d3_utils.py:237-246          scale_pts             (all n^2 pairwise distances)
    lib/d3_utils.py:144-148          rot_diff_rad / rot_diff_degree
    evaluation/parallel_ancsh_pose.py:20-33     ransac
    evaluation/parallel_ancsh_pose.py:35-54     single_transformation_{estimator,verifier}
    evaluation/parallel_ancsh_pose.py:56-68     objective_eval
    evaluation/parallel_ancsh_pose.py:106-194   joint_transformation_{estimator,verifier}
    evaluation/parallel_ancsh_pose.py:196-353   solver_ransac_nonlinear (per-cloud body -> solve_cloud)
    lib/aligning.py:580-622          estimateSimilarityUmeyama
    lib/aligning.py:17-32,485-507,540-547,88-103  estimateSimilarityTransform / getRANSACInliers /
                                      evaluateModel / set_config

The ONLY deliberate difference: the reference draws its 3-point samples from the unseeded global
numpy RNG inside the estimators (np.random.randint(n, size=3), :38, :110-111); here the draws come
from an explicit `SampleStream` so that the GPU path can consume the very same indices.  A stream
built with `SampleStream.from_seed` replays np.random.seed(s) + randint in the reference's call
order, so results are bit-identical to the imported reference run under that seed (checked by
tests/golden/gen_pose_golden.py, which is also what pins this oracle).

Third-party arithmetic (not under the reference tree): scipy.optimize.least_squares(method='lm')
= MINPACK lmdif (reference pins scipy==1.3.1, requirements.txt:151; here scipy 1.15.3 wraps the same
MINPACK routine), scipy Rotation from_dcm/as_dcm (removed in scipy>=1.6 -> from_matrix/as_matrix),
numpy.linalg.svd (LAPACK gesdd).
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as srot


# ------------------------------------------------------------------ lib/d3_utils.py
def rotate_points_with_rotvec(points, rot_vecs):                       # d3_utils.py:150-163
    theta = np.linalg.norm(rot_vecs, axis=1)[:, np.newaxis]
    with np.errstate(invalid='ignore'):
        v = rot_vecs / theta
        v = np.nan_to_num(v)
    dot = np.sum(points * v, axis=1)[:, np.newaxis]
    cos_theta = np.cos(theta)
    sin_theta = np.sin(theta)
    return cos_theta * points + sin_theta * np.cross(v, points) + dot * (1 - cos_theta) * v


def rotate_pts(source, target):                                        # d3_utils.py:206-220
    source = source - np.mean(source, 0, keepdims=True)
    target = target - np.mean(target, 0, keepdims=True)
    M = np.matmul(target.T, source)
    U, D, Vh = np.linalg.svd(M, full_matrices=True)
    d = (np.linalg.det(U) * np.linalg.det(Vh)) < 0.0
    if d:
        D[-1] = -D[-1]
        U[:, -1] = -U[:, -1]
    R = np.matmul(U, Vh)
    return R


def scale_pts(source, target):                                         # d3_utils.py:237-246
    pdist_s = source.reshape(source.shape[0], 1, 3) - source.reshape(1, source.shape[0], 3)
    A = np.sqrt(np.sum(pdist_s**2, 2)).reshape(-1)
    pdist_t = target.reshape(target.shape[0], 1, 3) - target.reshape(1, target.shape[0], 3)
    b = np.sqrt(np.sum(pdist_t**2, 2)).reshape(-1)
    scale = np.dot(A, b) / (np.dot(A, A) + 1e-6)
    return scale


def transform_pts(source, target):                                     # d3_utils.py:223-234
    source_centered = source - np.mean(source, 0, keepdims=True)
    target_centered = target - np.mean(target, 0, keepdims=True)
    rotation = rotate_pts(source_centered, target_centered)
    scale = scale_pts(source_centered, target_centered)
    translation = np.mean(target.T - scale * np.matmul(rotation, source.T), 1)
    return rotation, scale, translation


def rot_diff_rad(rot1, rot2):                                          # d3_utils.py:147-148
    return np.arccos((np.trace(np.matmul(rot1, rot2.T)) - 1) / 2) % (2 * np.pi)


def rot_diff_degree(rot1, rot2):                                       # d3_utils.py:144-145
    return rot_diff_rad(rot1, rot2) / np.pi * 180


# ------------------------------------------------------------------ sample streams
class SampleStream(object):
    """Pre-drawn 3-point sample indices, consumed in the reference's call order."""

    def __init__(self, draws):
        self.draws = list(draws)     # list of int arrays of shape (3,)
        self.pos = 0

    def next(self, n):
        d = self.draws[self.pos]
        self.pos += 1
        assert d.max() < n, "sample stream was drawn for a different point count"
        return d

    @staticmethod
    def from_seed(seed, plan):
        """plan: list of point counts in call order; replays np.random.seed(seed); randint(n, size=3)."""
        rs = np.random.RandomState(seed)
        return SampleStream([rs.randint(n, size=3) for n in plan])


def stage_a_plan(n, niter):
    return [n] * niter


def stage_b_plan(n0, n1, niter):
    return [n0, n1] * niter


# ------------------------------------------------------------------ evaluation/parallel_ancsh_pose.py
def ransac(dataset, model_estimator, model_verifier, inlier_th, niter, stream, info=None):   # :20-33
    best_model = None
    best_score = -np.inf
    best_inliers = None
    best_iter = -1
    for i in range(niter):
        cur_model = model_estimator(dataset, stream=stream)
        cur_score, cur_inliers = model_verifier(dataset, cur_model, inlier_th)
        if cur_score > best_score:                                      # strict: earliest iteration wins ties
            best_model = cur_model
            best_inliers = cur_inliers
            best_score = cur_score
            best_iter = i
    if info is not None:
        info.update(best_iter=best_iter, best_score=best_score, hyp_model=best_model)
    best_model = model_estimator(dataset, best_inliers)
    return best_model, best_inliers


def single_transformation_estimator(dataset, best_inliers=None, stream=None):               # :35-46
    if best_inliers is None:
        sample_idx = stream.next(dataset['nsource'])
    else:
        sample_idx = best_inliers
    rotation, scale, translation = transform_pts(dataset['source'][sample_idx, :], dataset['target'][sample_idx, :])
    return dict(rotation=rotation, scale=scale, translation=translation)


def single_transformation_verifier(dataset, model, inlier_th):                               # :48-54
    res = dataset['target'].T - model['scale'] * np.matmul(model['rotation'], dataset['source'].T) - model['translation'].reshape((3, 1))
    inliers = np.sqrt(np.sum(res**2, 0)) < inlier_th
    score = np.sum(inliers)
    return score, inliers


def objective_eval(params, x0, y0, x1, y1, joints, isweight=True):                           # :56-68
    rotvec0 = params[:3].reshape((1, 3))
    rotvec1 = params[3:].reshape((1, 3))
    res0 = y0 - rotate_points_with_rotvec(x0, rotvec0)
    res1 = y1 - rotate_points_with_rotvec(x1, rotvec1)
    res_joint = rotate_points_with_rotvec(joints, rotvec0) - rotate_points_with_rotvec(joints, rotvec1)
    if isweight:
        res0 /= x0.shape[0]
        res1 /= x1.shape[0]
        res_joint /= joints.shape[0]
    return np.concatenate((res0, res1, res_joint), 0).ravel()


def joint_transformation_estimator(dataset, best_inliers=None, stream=None, lm_log=None):    # :106-184 (revolute)
    if best_inliers is None:
        sample_idx0 = stream.next(dataset['nsource0'])
        sample_idx1 = stream.next(dataset['nsource1'])
    else:
        sample_idx0 = best_inliers[0]
        sample_idx1 = best_inliers[1]
    source0 = dataset['source0'][sample_idx0, :]
    target0 = dataset['target0'][sample_idx0, :]
    source1 = dataset['source1'][sample_idx1, :]
    target1 = dataset['target1'][sample_idx1, :]
    scale0 = scale_pts(source0, target0)
    scale1 = scale_pts(source1, target1)
    scale0_inv = scale_pts(target0, source0)
    scale1_inv = scale_pts(target1, source1)

    target0_scaled_centered = scale0_inv * target0
    target0_scaled_centered -= np.mean(target0_scaled_centered, 0, keepdims=True)
    source0_centered = source0 - np.mean(source0, 0, keepdims=True)
    target1_scaled_centered = scale1_inv * target1
    target1_scaled_centered -= np.mean(target1_scaled_centered, 0, keepdims=True)
    source1_centered = source1 - np.mean(source1, 0, keepdims=True)

    nj = np.min((source0.shape[0], source1.shape[0]))
    joint_points0 = np.ones_like(np.linspace(0, 1, num=nj + 1)[1:].reshape((-1, 1))) * dataset['joint_direction'].reshape((1, 3))

    R0 = rotate_pts(source0_centered, target0_scaled_centered)
    R1 = rotate_pts(source1_centered, target1_scaled_centered)
    rotvec0 = srot.from_matrix(R0).as_rotvec()          # reference: from_dcm (:147)
    rotvec1 = srot.from_matrix(R1).as_rotvec()
    x0 = np.hstack((rotvec0, rotvec1))
    res = least_squares(objective_eval, x0, verbose=0, ftol=1e-4, method='lm',
                        args=(source0_centered, target0_scaled_centered, source1_centered, target1_scaled_centered,
                              joint_points0, False))
    if lm_log is not None:
        lm_log.append(dict(x0=x0, x=res.x.copy(), nfev=res.nfev, status=res.status, cost=res.cost))
    R0 = srot.from_rotvec(res.x[:3]).as_matrix()        # reference: as_dcm (:156)
    R1 = srot.from_rotvec(res.x[3:]).as_matrix()
    translation0 = np.mean(target0.T - scale0 * np.matmul(R0, source0.T), 1)
    translation1 = np.mean(target1.T - scale1 * np.matmul(R1, source1.T), 1)
    return dict(rotation0=R0, scale0=scale0, translation0=translation0,
                rotation1=R1, scale1=scale1, translation1=translation1)


def joint_transformation_verifier(dataset, model, inlier_th):                                # :186-194
    res0 = dataset['target0'].T - model['scale0'] * np.matmul(model['rotation0'], dataset['source0'].T) - model['translation0'].reshape((3, 1))
    inliers0 = np.sqrt(np.sum(res0**2, 0)) < inlier_th
    res1 = dataset['target1'].T - model['scale1'] * np.matmul(model['rotation1'], dataset['source1'].T) - model['translation1'].reshape((3, 1))
    inliers1 = np.sqrt(np.sum(res1**2, 0)) < inlier_th
    score = (np.sum(inliers0) / res0.shape[0] + np.sum(inliers1) / res1.shape[0]) / 2       # res.shape[0] == 3 (sic)
    return score, [inliers0, inliers1]


def solve_cloud(P, nocs_pred, mask_pred, joint_axis_per_point, joint_cls_gt, num_parts, streams_a, streams_b,
                inlier_th=0.1, niter_a=10000, niter_b=200, info=None):
    """Per-cloud body of solver_ransac_nonlinear (:238-341) without file IO / GT error bookkeeping.

    P (N,3); nocs_pred (N,3K) and mask_pred (N,K) from the (baseline) part-NOCS network;
    joint_axis_per_point (N,3), joint_cls_gt (N,) from the ANCSH file.
    streams_a[j], streams_b[j-1]: SampleStream per part / per joint.
    Returns {'baseline': [(R,s,t)]*K, 'nonlinear': [(R,s,t)]*K, 'inliers_a': [...], 'inliers_b': [...]}.
    """
    cls_per_pt_pred = np.argmax(mask_pred, axis=1)                                            # :238
    partidx = [np.where(cls_per_pt_pred == j)[0] for j in range(num_parts)]
    joint_idx_list_gt = [np.where(joint_cls_gt == j)[0] for j in range(1, num_parts)]
    out = dict(baseline=[], nonlinear=[None] * num_parts, inliers_a=[], inliers_b=[], info_a=[], info_b=[])
    for j in range(num_parts):                                                                # stage A :258-285
        dataset = dict(source=nocs_pred[partidx[j], 3 * j:3 * (j + 1)], target=P[partidx[j], :3])
        dataset['nsource'] = dataset['source'].shape[0]
        inf = {}
        best_model, best_inliers = ransac(dataset, single_transformation_estimator, single_transformation_verifier,
                                          inlier_th, niter_a, streams_a[j], inf)
        out['baseline'].append((best_model['rotation'], best_model['scale'], best_model['translation']))
        out['inliers_a'].append(best_inliers)
        out['info_a'].append(inf)
    for j in range(1, num_parts):                                                             # stage B :287-341
        dataset = dict(source0=nocs_pred[partidx[0], :3], target0=P[partidx[0], :3],
                       source1=nocs_pred[partidx[j], 3 * j:3 * (j + 1)], target1=P[partidx[j], :3])
        dataset['nsource0'] = dataset['source0'].shape[0]
        dataset['nsource1'] = dataset['source1'].shape[0]
        dataset['joint_direction'] = np.median(joint_axis_per_point[joint_idx_list_gt[j - 1], :], 0)   # :295
        inf = {}
        best_model, best_inliers = ransac(dataset, joint_transformation_estimator, joint_transformation_verifier,
                                          inlier_th, niter_b, streams_b[j - 1], inf)
        if j == 1:
            out['nonlinear'][0] = (best_model['rotation0'], best_model['scale0'], best_model['translation0'])
        out['nonlinear'][j] = (best_model['rotation1'], best_model['scale1'], best_model['translation1'])
        out['inliers_b'].append(best_inliers)
        out['info_b'].append(inf)
    if info is not None:
        info.update(partidx=partidx)
    return out


# ------------------------------------------------------------------ lib/aligning.py
def estimateSimilarityUmeyama(SourceHom, TargetHom, rt_pre=None):                            # aligning.py:580-622
    SourceCentroid = np.mean(SourceHom[:3, :], axis=1)
    TargetCentroid = np.mean(TargetHom[:3, :], axis=1)
    nPoints = SourceHom.shape[1]
    CenteredSource = SourceHom[:3, :] - np.tile(SourceCentroid, (nPoints, 1)).transpose()
    CenteredTarget = TargetHom[:3, :] - np.tile(TargetCentroid, (nPoints, 1)).transpose()
    CovMatrix = np.matmul(CenteredTarget, np.transpose(CenteredSource)) / nPoints
    if np.isnan(CovMatrix).any():
        raise RuntimeError('There are NANs in the input.')
    U, D, Vh = np.linalg.svd(CovMatrix, full_matrices=True)
    d = (np.linalg.det(U) * np.linalg.det(Vh)) < 0.0
    if d:
        D[-1] = -D[-1]
        U[:, -1] = -U[:, -1]
    if rt_pre is not None:
        Rotation = rt_pre[:3, :3].T
    else:
        Rotation = np.matmul(U, Vh).T                       # "Transpose is the one that works"
    varP = np.var(SourceHom[:3, :], axis=1).sum()
    ScaleFact = 1 / varP * np.sum(D)
    Scales = np.array([ScaleFact, ScaleFact, ScaleFact])
    ScaleMatrix = np.diag(Scales)
    Translation = TargetHom[:3, :].mean(axis=1) - SourceHom[:3, :].mean(axis=1).dot(ScaleFact * Rotation)
    OutTransform = np.identity(4)
    OutTransform[:3, :3] = ScaleMatrix @ Rotation.T
    OutTransform[:3, 3] = Translation
    return Scales, Rotation, Translation, OutTransform


def set_config(source, target):                                                              # aligning.py:88-103
    SourceHom = np.transpose(np.hstack([source, np.ones([source.shape[0], 1])]))
    TargetHom = np.transpose(np.hstack([target, np.ones([target.shape[0], 1])]))
    TargetNorm = np.mean(np.linalg.norm(target, axis=1))
    SourceNorm = np.mean(np.linalg.norm(source, axis=1))
    RatioTS = (TargetNorm / SourceNorm)
    RatioST = (SourceNorm / TargetNorm)
    PassT = RatioST if (RatioST > RatioTS) else RatioTS
    StopT = PassT / 100
    return SourceHom, TargetHom, PassT, StopT


def evaluateModel(OutTransform, SourceHom, TargetHom, PassThreshold):                        # aligning.py:540-547
    Diff = TargetHom - np.matmul(OutTransform, SourceHom)
    ResidualVec = np.linalg.norm(Diff[:3, :], axis=0)
    Residual = np.linalg.norm(ResidualVec)
    InlierIdx = np.where(ResidualVec < PassThreshold)
    nInliers = np.count_nonzero(InlierIdx)      # (sic) counts non-zero INDICES: point 0 never counts
    InlierRatio = nInliers / SourceHom.shape[1]
    return Residual, InlierRatio, InlierIdx[0]


def getRANSACInliers(SourceHom, TargetHom, draws, MaxIterations=100, PassThreshold=200, StopThreshold=1):   # :485-507
    BestResidual = 1e10
    BestInlierRatio = 0
    BestInlierIdx = np.arange(SourceHom.shape[1])
    for i in range(0, MaxIterations):
        RandIdx = draws[i]                                  # reference: np.random.randint(n, size=5)
        _s, _r, _t, OutTransform = estimateSimilarityUmeyama(SourceHom[:, RandIdx], TargetHom[:, RandIdx])
        Residual, InlierRatio, InlierIdx = evaluateModel(OutTransform, SourceHom, TargetHom, PassThreshold)
        if InlierRatio > BestInlierRatio:
            BestResidual = Residual
            BestInlierRatio = InlierRatio
            BestInlierIdx = InlierIdx
        if BestResidual < StopThreshold:
            break
    return SourceHom[:, BestInlierIdx], TargetHom[:, BestInlierIdx], BestInlierRatio


def estimateSimilarityTransform(source, target, draws):                                      # aligning.py:17-32
    SourceHom, TargetHom, PassT, StopT = set_config(source, target)
    SourceInliersHom, TargetInliersHom, BestInlierRatio = getRANSACInliers(
        SourceHom, TargetHom, draws, MaxIterations=100, PassThreshold=PassT, StopThreshold=StopT)
    if BestInlierRatio < 0.1:
        return None, None, None, None
    return estimateSimilarityUmeyama(SourceInliersHom, TargetInliersHom)


def compose_rt(rotation, translation):                                                       # evaluation/compute_gt_pose.py:14-19
    aligned_RT = np.zeros((4, 4), dtype=np.float32)
    aligned_RT[:3, :3] = rotation.transpose()
    aligned_RT[:3, 3] = translation
    aligned_RT[3, 3] = 1
    return aligned_RT
