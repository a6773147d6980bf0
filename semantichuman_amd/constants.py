"""Configuration VALUES of the reference that the hot path reads (reference configure/cfgs.py and
configure/traincfg.yaml).  The reference reaches them through a global yacs `cfg`; here they are
plain Python data that callers may override through keyword arguments."""

# configure/traincfg.yaml:55 (overrides cfgs.py:22-24): bones as (child, parent[, parent2]) joint ids
NEWSKL_LIST = [[0, 1], [0, 2], [0, 6], [1, 4], [2, 5], [6, 9], [4, 7], [5, 8], [9, 12], [9, 16], [9, 17], [7, 10], [8, 11],
               [12, 15], [16, 18], [17, 19], [18, 20], [19, 21], [20, 22], [21, 23], [20, 24], [21, 25], [20, 26], [21, 27],
               [15, 28], [15, 29], [15, 30], [7, 31], [8, 32], [7, 33], [8, 34]]
# configure/traincfg.yaml:56: joints (after dropping 3, 13, 14) that describe each of the 17 parts
KPS_INDEX_LIST = [[12, 25, 26, 27], [12, 11], [11, 8], [5, 0], [0, 1, 2], [1, 3], [3, 6], [6, 9, 28, 30], [2, 4], [4, 7],
                  [7, 10, 29, 31], [13, 15], [15, 17], [17, 19, 21, 23], [14, 16], [16, 18], [18, 20, 22, 24]]
# configure/cfgs.py:19-21: bone of each part (in the full 24+ joint numbering), used by the angle weights
SKL_LIST = [[15, 12], [15, 12], [12, 9], [6, 0], [0, 1, 2], [1, 4], [4, 7], [7, 10], [2, 5], [5, 8], [8, 11], [16, 18],
            [18, 20], [20, 22], [17, 19], [19, 21], [21, 23]]
PART_LIST = ['head', 'neck', 'chest', 'abdomen', 'hip', 'left_ham', 'left_shank', 'left_feet', 'right_ham', 'right_shank',
             'right_feet', 'left_arm', 'left_forearm', 'left_hand', 'right_arm', 'right_forearm', 'right_hand']
LEAF_PART_LIST = ['head', 'left_feet', 'right_feet', 'left_hand', 'right_hand']
NOLEAF_PART_LIST = ['neck', 'chest', 'abdomen', 'hip', 'left_ham', 'left_shank', 'right_ham', 'right_shank', 'left_arm',
                    'left_forearm', 'right_arm', 'right_forearm']
MEASURE_PART_LIST = ['neck', 'chest', 'abdomen', 'hip', 'left_ham', 'left_shank', 'left_feet', 'right_ham', 'right_shank',
                     'right_feet', 'left_arm', 'left_forearm', 'left_hand', 'right_arm', 'right_forearm', 'right_hand']
KPS_DROPPED = [3, 13, 14]          # models.py:170-171, utils_SH.py:33-35

# configure/traincfg.yaml:2-10
FILTER_SIZES_ENC = [[3, 16, 32, 64, 128], [[], [], [], [], []]]
FILTER_SIZES_DEC = [[128, 64, 32, 32, 16], [[], [], [], [], 3]]
DS_FACTORS = [2, 2, 2, 2]
STEP_SIZES = [2, 2, 1, 1, 1]
DILATION = [2, 2, 1, 1, 1]
NZ = 256
PART_SHAPE_LATENT_SIZE = 8
PART_KPS_LATENT_SIZE = 8
# configure/cfgs.py:89-91, traincfg.yaml:40-52
LR, WEIGHT_DECAY, SCHEDULER = 1e-3, 5e-5, (True, 1, 0.99)
LOSS_WEIGHTS = dict(edgereg_w=1e-2, zpartreg_w=1e-2, vol_w=1e-2, interp_kps_w=1.0, interp_euc_w=1e-2, exc_kps_w=1.0,
                    exc_euc_w=1e-2)
W_MODE, W_THRESHOLD, W_PART_MODE, RELAT_FLAG = 'threshold', 0.8, '1/K', True


def kps_keep(newskl_list=NEWSKL_LIST):
    return [i for i in range(len(newskl_list) + 4) if i not in KPS_DROPPED]
