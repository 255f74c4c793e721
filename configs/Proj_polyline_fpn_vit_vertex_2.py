# Inference-relevant subset of the reference's config of the same name (values: SURVEY.md §A.1).
# The loader also reads the reference's own configs/Proj_*.py unchanged.
seed = 2021
view = False
number_lanes = 12
number_orients = 11
flip_label = False
is_gt_avai = False
net = dict(type='Detector1stage', head_type='row', loss_type='row_ce')
pcencoder = dict(type='PostProjector2', resnet='resnet34', pretrained=False,
                 replace_stride_with_dilation=[False, True, False], out_conv=True, in_channels=[64, 128, 256, -1])
featuremap_out_channel = 64
list_img_size_xy = [1152, 1152]
backbone = dict(type='VitSegNet', image_size=144, patch_h_size=8, patch_w_size=8, channels=64, dim=512, depth=3,
                heads=16, output_channels=8, expansion_factor=4, dim_head=64, dropout=0., emb_dropout=0.,
                is_with_shared_mlp=False, is_with_llm=False)
heads = dict(type='ColumnProposal2', dim_feat=8, row_size=144, dim_shared=100, num_prop=72, prop_width=2,
             prop_half_buff=4, dim_token=512, tr_depth=1, tr_heads=16, tr_dim_head=64, tr_mlp_dim=512,
             row_dim_token=96, row_tr_depth=1, row_tr_heads=12, row_tr_dim_head=8, row_tr_mlp_dim=144,
             endp_mode='endp_est', cls_exp=True)
proposal_obj_thre = 0.3
exist_thre = 0.2
coor_thre = 0.2
endp_thre = 0.08
show_result = False
view_detail = False
dataset_type = 'LaserLaneProposal'
vit_seg = True
column_att = False
column_transformer_decoder = False
spatial_att = True
cls_smooth = False

# entry-point contract (load_config_and_runner / Runner.infer_*: baseline/engine/runner.py:57-66, :690-697)
log_dir = './logs'
distributed = False
batch_size = 6
validate_buffer = 10
gt_downsample_ratio = 8
workers = 12
dataset_path = './data/LaserLane/TrainValAll'
data_split_file = 'data_split-shuffle.json'
dataset_color_augment = False
dataset = dict(
    train=dict(type=dataset_type, data_root=dataset_path, data_split_file=data_split_file, mode='train'),
    val=dict(type=dataset_type, data_root=dataset_path, data_split_file=data_split_file, mode='val'),
    test=dict(type=dataset_type, data_root=dataset_path, data_split_file=data_split_file, mode='test'),
)
