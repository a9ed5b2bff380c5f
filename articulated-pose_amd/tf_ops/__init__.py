from .tf_sampling import farthest_point_sample, gather_point, prob_sample  # noqa: F401
from .tf_grouping import (query_ball_point, group_point, query_ball_group_xyz, query_ball_group_xyz_multi, query_ball_point_multi,  # noqa: F401
                          group_point_multi, select_top_k, knn_point)
from .tf_interpolate import three_nn, three_interpolate  # noqa: F401
