"""Net assembly behind the reference's NET registry names ``Detector1stage`` and ``Segmentor``.

Drop-in for baseline/models/net/detector1stage.py:12-67 and segmentor.py:15-41 (eval path only; training
losses are out of scope).  `Detector1stage` keeps the dead `conv1 = Conv2d(144,144,3,3)` parameter
(detector1stage.py:18) for strict checkpoint loading.
"""
import torch
import torch.nn as nn

from . import ops, trace
from .registry import NET, build_pcencoder, build_backbone, build_heads


@NET.register_module
class Detector1stage(nn.Module):
    def __init__(self, head_type='seg', loss_type='row_ce', cfg=None):
        super().__init__()
        self.cfg = cfg
        self.conv1 = nn.Conv2d(144, 144, 3, 3)
        self.pcencoder = build_pcencoder(cfg)
        self.backbone = build_backbone(cfg)
        self.heads = build_heads(cfg)
        self.head_type, self.loss_type = head_type, loss_type

    def forward_raw(self, batch):
        """pcencoder -> backbone -> heads, raw outputs (detector1stage.py:28-51)."""
        fused = hasattr(self.pcencoder, 'fpn') and self.cfg.heads.type == 'ColumnProposal2'
        col = None
        if fused:   # FPN writes fea_up straight into channels 8..15 of the head's concat buffer
            proj = batch['proj']            # [B,3,H,W] f32 (the reference's tensor) or [B,H,W,3] uint8 (rasteriser / PNG output)
            B, H, W = (proj.shape[0], proj.shape[1], proj.shape[2]) if proj.dtype == torch.uint8 else (proj.shape[0], proj.shape[2], proj.shape[3])
            col = ops.new_act(B, 16, H // 4, W // 4, proj.device)
            with trace.stage('pcencoder'):
                fea, fea_up, bi_seg, endp_est = self.pcencoder.fpn(proj, fea_up_out=col[:, 8:16])
        else:
            with trace.stage('pcencoder'):
                fea, fea_up, bi_seg, endp_est = self.pcencoder(batch)
        if self.cfg.vit_seg == True:   # noqa: E712  (the reference compares with == True)
            with trace.stage('backbone'):
                fea = self.backbone(fea)
        with trace.stage('heads'):
            if self.cfg.heads.type == 'RowSharNotReducRef':
                out = self.heads(fea)
            else:
                out = self.heads(fea, fea_up, endp_est, col=col) if fused else self.heads(fea, fea_up, endp_est)
        out['semantic_seg'] = bi_seg
        out['endp_est'] = endp_est
        return out

    def forward(self, batch, is_get_features=False, stack_local_global_features=False):
        if self.training:
            raise NotImplementedError('lanemapping_amd implements the inference hot path only (call .eval())')
        if is_get_features:
            raise NotImplementedError('is_get_features is not on the hot path')
        output = {}
        with torch.no_grad():
            out = self.forward_raw(batch)
            output.update(self.heads.get_exist_coor_endp_dict(out))
            output['lane_maps'] = self.heads.get_lane_map_numpy_with_label(
                output, batch, is_flip=self.cfg.flip_label, is_img=self.cfg.view, is_get_1_stage_result=False,
                is_gt_avai=self.cfg.is_gt_avai)
            if self.cfg.show_result:
                output['pred_maps'] = self.heads.get_lane_map_on_source_image(output, batch)
        return output


@NET.register_module
class Segmentor(nn.Module):
    def __init__(self, head_type='seg', loss_type='ce', cfg=None):
        super().__init__()
        self.cfg = cfg
        self.pcencoder = build_pcencoder(cfg)
        self.head_type, self.loss_type = head_type, loss_type

    def forward(self, batch):
        if self.training:
            raise NotImplementedError('lanemapping_amd implements the inference hot path only (call .eval())')
        with torch.no_grad():
            _, _, bi_seg, endp_est = self.pcencoder(batch)
            pred = {'seg': bi_seg, 'endp': endp_est}
            return dict(self.pcencoder.infer_validate(pred, seg_thre=self.cfg.seg_thre, endp_thre=self.cfg.endp_thre))
