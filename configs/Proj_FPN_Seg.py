# Inference-relevant subset of the reference's Segmentor config (values: SURVEY.md §A.1).
seed = 2021
number_lanes = 12
number_orients = 11
flip_label = False
net = dict(type='Segmentor', head_type='seg', loss_type='ce')
pcencoder = dict(type='PostProjector2', resnet='resnet34', pretrained=False,
                 replace_stride_with_dilation=[False, True, False], out_conv=True, in_channels=[64, 128, 256, -1])
featuremap_out_channel = 64
list_img_size_xy = [1152, 1152]
view = False
seg_thre = 0.1
endp_thre = 0.1
dataset_type = 'LaserLane'

# entry-point contract (load_config_and_runner / Runner.infer_*: baseline/engine/runner.py:57-66, :690-697)
log_dir = './logs'
distributed = False
batch_size = 6
workers = 12
dataset_path = './data/LaserLane/All'
dataset = dict(
    train=dict(type=dataset_type, data_root=dataset_path, mode='train'),
    val=dict(type=dataset_type, data_root=dataset_path, mode='val'),
    test=dict(type=dataset_type, data_root=dataset_path, mode='test'),
)
