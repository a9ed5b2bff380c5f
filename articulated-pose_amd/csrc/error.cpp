// error.cpp -- per-thread error string + ABI version of libancsh_hip.so.
#include "common.h"

namespace ancsh {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ancsh

extern "C" int ancsh_abi_version(void) { return 7; }   // 7: ancsh_hbm_copy (a bench yardstick, not an operator) moved out of the library into tools/microbench; 6: + the split-16 experiment's grouped / tail / F16x2 entry points (ancsh_*_bf16x3_grouped, ancsh_mlp_chain_grouped_fp_bf16x3, ancsh_*_f16x2*), ancsh_pose_poison_records; tie_stats[1] of ancsh_ransac_single_rec redefined (sign = degenerate winner); 5: + the mid-section chains (ancsh_sa3_chain_grouped, ancsh_fp_single_source_init, ancsh_fp1_chain_grouped, ancsh_fp2_chain_grouped), ancsh_mlp_chain_grouped_fp, ancsh_ransac_single_rec / ancsh_ransac_joint_rec, ancsh_last_ball_query_schedule; 4: + ancsh_joint_params, ancsh_part_extents, ancsh_query_ball_group_xyz_multi; LM_AUTO = THROUGHPUT (3: grouped launches, ancsh_ransac_single_ex, ancsh_three_nn_weights; additions only)
extern "C" const char *ancsh_last_error(void) { return ancsh::g_err; }
