# Inference-relevant subset of the reference's K-Lane RowRef config (values: SURVEY.md §A.1).  The reference file lacks
# vit_seg / is_gt_avai / number_orients; lanemapping_amd.config.apply_inference_defaults injects them.
seed = 2021
view = False
number_lanes = 12
flip_label = False
net = dict(type='Detector1stage', head_type='row', loss_type='row_ce')
pcencoder = dict(type='PostProjector2', resnet='resnet34', pretrained=False,
                 replace_stride_with_dilation=[False, True, False], out_conv=True, in_channels=[64, 128, 256, -1])
featuremap_out_channel = 64
list_img_size_xy = [1152, 1152]
backbone = dict(type='VitSegNet', image_size=144, patch_h_size=8, patch_w_size=8, channels=64, dim=512, depth=3, heads=16,
                output_channels=1024, expansion_factor=4, dim_head=64, dropout=0., emb_dropout=0., is_with_shared_mlp=False)
heads = dict(type='RowSharNotReducRef', dim_feat=8, row_size=144, dim_shared=512, lambda_cls=1., thr_ext=0.3, off_grid=2,
             dim_token=1024, tr_depth=1, tr_heads=16, tr_dim_head=64, tr_mlp_dim=2048, tr_dropout=0., tr_emb_dropout=0.,
             is_reuse_same_network=False)
conf_thr = 0.5
show_result = False
dataset_type = 'LaserLane'

# entry-point contract (load_config_and_runner / Runner.infer_*: baseline/engine/runner.py:57-66, :690-697)
log_dir = './logs'
distributed = False
batch_size = 6
workers = 12
dataset_path = './../All'
dataset = dict(
    train=dict(type=dataset_type, data_root=dataset_path, mode='train'),
    test=dict(type=dataset_type, data_root=dataset_path, mode='test'),
)
