#!/bin/bash
# The reference's evaluation.sh on this build: the same five steps with the same flags (ITEM / DOMAIN / BASE as environment overrides).
# Every step reads and writes the reference's files under $BASE/results (prediction records .h5 or .npz, pose / ground-truth pickles).
# compute_gt_pose runs for both NOCS types: steps 3-5 read both ground-truth pickles (eval_pose_err.py:66-67).
set -e
ITEM=${ITEM:-eyeglasses}
DOMAIN=${DOMAIN:-unseen}
BASE=${BASE:-${ANCSH_BASE_PATH:-$(pwd)}}
cd "$(dirname "$0")"
python -m articulated_pose_amd.compute_gt_pose --item=$ITEM --domain=$DOMAIN --nocs=ANCSH --save --base_path "$BASE"
python -m articulated_pose_amd.compute_gt_pose --item=$ITEM --domain=$DOMAIN --nocs=NAOCS --save --base_path "$BASE"

# run our processing over test group (one worker rank per visible MI355X)
python -m articulated_pose_amd.pose_multi_process --item=$ITEM --domain=$DOMAIN --base_path "$BASE"

# pose & relative joint rotation
python -m articulated_pose_amd.eval_pose_err --item=$ITEM --domain=$DOMAIN --nocs=ANCSH --base_path "$BASE"

# 3d miou estimation
python -m articulated_pose_amd.compute_miou --item=$ITEM --domain=$DOMAIN --nocs=ANCSH --base_path "$BASE"

# performance on joint estimations
python -m articulated_pose_amd.eval_joint_params --item=$ITEM --domain=$DOMAIN --nocs=ANCSH --base_path "$BASE"
