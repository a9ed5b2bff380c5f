"""Read-only category table the hot path needs from the reference's global_info.py (:15-180):
num_parts, experiment ids, unseen/special instance lists.  `base_path` is overridable (the reference
hard-codes '/work/cascades/lxiaol9/6DPOSE', global_info.py:192)."""
import collections
import os

DatasetInfo = collections.namedtuple('DatasetInfo', ['num_parts', 'test_list', 'spec_list', 'exp', 'baseline'])

_DATASETS = dict(
    eyeglasses=DatasetInfo(3, ['0007', '0016', '0036'], ['0006'], '3.9', '3.91'),
    oven=DatasetInfo(2, ['0003', '0016', '0029'], ['0006', '0015', '0035', '0038'], '3.0', '3.01'),
    laptop=DatasetInfo(2, ['0004', '0008', '0069'], ['0003', '0006', '0041', '0080', '0081'], '3.6', '3.61'),
    washing_machine=DatasetInfo(2, [], [], '3.1', '3.11'),
    drawer=DatasetInfo(4, [], [], '3.3', '3.31'),
)


class global_info(object):
    def __init__(self, base_path=None):
        self.datasets = _DATASETS
        self.base_path = base_path or os.environ.get('ANCSH_BASE_PATH', os.getcwd())


def get_test_group(all_test_h5, unseen_instances, domain='seen', spec_instances=[], category=None):
    """lib/data_utils.py:908-934: unseen = held-out instances, every 5th frame; seen = every 3rd articulation.
    Accepts '.h5' (reference) and '.npz' (this build's fallback container) record files."""
    seen_test_h5, unseen_test_h5 = [], []
    seen_arti_select = [str(x) for x in range(0, 31, 3)]
    unseen_frame_select = [str(x) for x in range(0, 30, 5)]
    for test_h5 in all_test_h5:
        if test_h5[0:4] in spec_instances or not (test_h5.endswith('.h5') or test_h5.endswith('.npz')):
            continue
        name_info = test_h5.split('.')[0].split('_')
        item, art_index, frame_order = name_info[0], name_info[1], name_info[2]
        if item in unseen_instances and frame_order in unseen_frame_select:
            unseen_test_h5.append(test_h5)
        elif item not in unseen_instances and art_index in seen_arti_select:
            seen_test_h5.append(test_h5)
    return seen_test_h5 if domain == 'seen' else unseen_test_h5
