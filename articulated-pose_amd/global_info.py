"""Read-only category table the hot path needs from the reference's global_info.py (:15-180):
num_parts, experiment ids, unseen/special instance lists.  `base_path` is overridable (the reference
hard-codes '/work/cascades/lxiaol9/6DPOSE', global_info.py:192)."""
import collections
import os

DatasetInfo = collections.namedtuple('DatasetInfo', ['num_parts', 'test_list', 'spec_list', 'exp', 'baseline', 'dataset_name', 'spec_map'],
                                     defaults=['shape2motion', None])

# part order of every drawer instance (global_info.py:170-177): spec_map[instance][k] = the URDF link that is part k
_DRAWER_ORDER = {i: [3, 0, 1, 2] for i in ('40453', '44962', '45132', '45290', '46123', '46130', '46334', '46440', '46462', '46537', '46544', '46641',
                                          '47178', '47183', '47296', '47233', '48010', '48253', '48517', '48740', '48876', '46230')}
_DRAWER_ORDER.update({'44853': [3, 1, 2, 0], '45135': [3, 1, 0, 2], '45427': [3, 2, 0, 1], '45756': [3, 1, 2, 0], '45841': [0, 1, 2, 3],
                      '46653': [0, 1, 2, 3], '46879': [3, 1, 2, 0], '47438': [3, 2, 1, 0], '47711': [0, 1, 2, 3], '48491': [0, 1, 2, 3]})

_DATASETS = dict(
    eyeglasses=DatasetInfo(3, ['0007', '0016', '0036'], ['0006'], '3.9', '3.91'),
    oven=DatasetInfo(2, ['0003', '0016', '0029'], ['0006', '0015', '0035', '0038'], '3.0', '3.01'),
    laptop=DatasetInfo(2, ['0004', '0008', '0069'], ['0003', '0006', '0041', '0080', '0081'], '3.6', '3.61'),
    washing_machine=DatasetInfo(2, ['0003', '0029'], ['0001', '0002', '0006', '0007', '0010', '0027', '0031', '0040', '0050', '0009', '0029', '0038',
                                                      '0039', '0041', '0046', '0052', '0058'], '3.1', '3.11'),
    drawer=DatasetInfo(4, ['46123', '45841', '46440'], [], '3.3', '3.31', 'sapien', _DRAWER_ORDER),
)


class global_info(object):
    def __init__(self, base_path=None, group_path=None):
        self.datasets = _DATASETS
        self.base_path = base_path or os.environ.get('ANCSH_BASE_PATH', os.getcwd())
        self.group_path = group_path or os.environ.get('ANCSH_GROUP_PATH', self.base_path)     # global_info.py:193: where the sapien set lives


_RECORD_SUFFIXES = ('.h5', '.npz')      # reference container / this build's fallback container


def _parse_record_name(filename):
    """'<instance>_<articulation>_<frame>.<ext>' (lib/dataset.py:78) -> (instance, articulation:int, frame:int) or None."""
    stem, dot, _ext = filename.partition('.')
    fields = stem.split('_')
    if not dot or len(fields) < 3 or not (fields[1].isdigit() and fields[2].isdigit()):
        return None
    if fields[1] != str(int(fields[1])) or fields[2] != str(int(fields[2])):
        return None                     # the reference compares the decimal strings: '05' is not frame 5
    return fields[0], int(fields[1]), int(fields[2])


def get_test_group(all_test_h5, unseen_instances, domain='seen', spec_instances=[], category=None):
    """Evaluation split of the record files, same selection as lib/data_utils.py:908-934:
      domain 'unseen': records of the held-out instances, frames 0, 5, ..., 25;
      domain 'seen'  : records of every other instance, articulations 0, 3, ..., 30;
    instances in `spec_instances` (matched on the first four characters) are dropped.  Input order is kept."""
    held_out, special = frozenset(unseen_instances), frozenset(spec_instances)
    want_unseen = domain != 'seen'

    def selected(filename):
        if filename[:4] in special or not filename.endswith(_RECORD_SUFFIXES):
            return False
        parsed = _parse_record_name(filename)
        if parsed is None:
            return False
        instance, articulation, frame = parsed
        if instance in held_out:
            return want_unseen and frame % 5 == 0 and frame < 30
        return (not want_unseen) and articulation % 3 == 0 and articulation <= 30

    return [f for f in all_test_h5 if selected(f)]
