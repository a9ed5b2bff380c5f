"""Pose-fit half on the MI355X: counterparts of evaluation/parallel_ancsh_pose.py, lib/d3_utils.py and
lib/aligning.py (the reference runs these in numpy/scipy, one worker process per CPU core)."""
from .parallel_ancsh_pose import PoseSolver, ransac_single_batch, ransac_joint_batch, solver_ransac_nonlinear  # noqa: F401
from .aligning import (estimateSimilarityUmeyama, umeyama_batch, estimateSimilarityTransform,  # noqa: F401
                       estimate_similarity_transform_batch)
