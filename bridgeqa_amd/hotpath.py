"""The slice of `ScanQA.forward` that IS the hot path (reference models/qa_module.py:438-479 for the
detector branch and :611-668 for the BLIP branch), with the reference's attribute names so its
state-dict keys are a subset of ScanQA's: detection_backbone.*, voting_net.*, proposal_net.*,
object_feat_linear.*, blip_model.*.

Round 2 widened it to the rest of the VQA branch: the language-classification head and the
reference-object head that follow the fusion (qa_module.py:735-754 over the modules of :234-249: lang_cls, object_cls,
linear_blip_to_object, dec_list_qo; enc_list_o exists for state-dict parity only -- the reference never calls it),
switched on by the reference's own flags use_lang_cls / use_reference (default off, as in ScanQA.__init__).  The
non-BLIP branch (LSTM language module, MCAN fusion backbone, AttFlat heads: qa_module.py:496-590) stays out.
"""
import numpy as np
import torch
import torch.nn as nn

from .backbone_module import Pointnet2Backbone
from .mcan_module import SA, SGA
from .proposal_module import ProposalModule
from .voting_module import VotingModule


def build_qa_heads(module, hidden_size, blip_enc_size, num_object_class, mcan_num_layers=2, mcan_num_heads=8,
                   mcan_pdrop=0.1):
    """registers, on `module`, the heads ScanQA.__init__ creates under use_blip (qa_module.py:223-249), same names and
    registration order: enc_list_o, lang_cls, object_cls, linear_blip_to_object, dec_list_qo"""
    module.enc_list_o = nn.ModuleList([SA(hidden_size, mcan_num_heads, mcan_pdrop) for _ in range(mcan_num_layers)])
    module.lang_cls = nn.Sequential(nn.Linear(blip_enc_size, hidden_size), nn.GELU(), nn.Dropout(0.1),
                                    nn.Linear(hidden_size, num_object_class))
    module.object_cls = nn.Sequential(nn.Linear(hidden_size, hidden_size), nn.GELU(), nn.Dropout(0.1),
                                      nn.Linear(hidden_size, 1))
    module.linear_blip_to_object = nn.Linear(blip_enc_size, hidden_size)
    module.dec_list_qo = nn.ModuleList([SGA(hidden_size, mcan_num_heads, mcan_pdrop) for _ in range(mcan_num_layers)])
    return module


def qa_heads_forward(module, data_dict, object_feat, object_mask, fused_feat, fused_mask, use_lang_cls, use_reference):
    """qa_module.py:735-754.  object_feat (B,K,hidden); object_mask (B,1,1,K) bool, True = NOT an object; fused_feat
    (B,L,blip width) and fused_mask (B,L), 1 = real token, from BLIP_VQA3D.  Writes lang_scores (B, classes) and
    cluster_ref (B,K).  Kept as the reference has it: the decoder's self-attention receives ~object_mask, i.e. it hides
    the VALID proposals from each other (qa_module.py:750) -- parity first; flagged here, not "fixed"."""
    fused_feat = fused_feat.float()
    if use_lang_cls:
        data_dict["lang_scores"] = module.lang_cls(fused_feat[:, 0, :])
    if use_reference:
        y = module.linear_blip_to_object(fused_feat)
        y_mask = fused_mask.unsqueeze(1).unsqueeze(2).bool()
        x = object_feat.float()
        for dec in module.dec_list_qo:
            x = dec(x, y, ~object_mask, ~y_mask, att_pdrop=None, att_drop_topk=None)
        conf = x * data_dict["objectness_scores"].max(2)[1].float().unsqueeze(2)
        data_dict["cluster_ref"] = module.object_cls(conf).squeeze(-1)
    return data_dict


class ScanQAHotPath(nn.Module):
    def __init__(self, input_feature_dim=132, num_proposal=256, vote_factor=1, seed_feat_dim=256, proposal_size=128,
                 vote_radius=0.3, vote_nsample=16, hidden_size=256, num_class=18, num_heading_bin=1,
                 num_size_cluster=18, mean_size_arr=None, use_blip=True, blip_kwargs=None, use_lang_cls=False,
                 use_reference=False, qa_heads=None, mcan_num_layers=2, mcan_num_heads=8, mcan_pdrop=0.1):
        """use_lang_cls / use_reference: ScanQA's flags for the two heads after the fusion; qa_heads: create their
        modules (default: iff one of the flags is set; ScanQA itself always creates them under use_blip -- pass True to
        load a reference checkpoint with strict=True)"""
        super().__init__()
        if mean_size_arr is None:
            mean_size_arr = np.ones((num_size_cluster, 3))
        self.detection_backbone = Pointnet2Backbone(input_feature_dim=input_feature_dim, seed_feat_dim=seed_feat_dim)
        self.voting_net = VotingModule(vote_factor, seed_feat_dim)
        self.proposal_net = ProposalModule(num_class, num_heading_bin, num_size_cluster, mean_size_arr, num_proposal,
                                           "vote_fps", seed_feat_dim=seed_feat_dim, proposal_size=proposal_size,
                                           radius=vote_radius, nsample=vote_nsample)
        self.object_feat_linear = nn.Sequential(nn.Linear(proposal_size, hidden_size), nn.GELU())  # qa_module.py:219-221
        self.use_blip = use_blip
        if use_blip:
            from .blip_vqa_3d import BLIP_VQA3D
            kw = dict(num_answers=10, use_text_decoder=True, share_decoder=True, scene_size=hidden_size,
                      scene_feature_position="paralleltwin")
            kw.update(blip_kwargs or {})
            self.blip_model = BLIP_VQA3D(**kw)
            self.use_lang_cls, self.use_reference = use_lang_cls, use_reference
            if qa_heads if qa_heads is not None else (use_lang_cls or use_reference):
                build_qa_heads(self, hidden_size, self.blip_model.text_encoder.config.hidden_size, num_class,
                               mcan_num_layers, mcan_num_heads, mcan_pdrop)

    def detect(self, data_dict):
        data_dict = self.detection_backbone(data_dict)
        xyz, features = data_dict["fp2_xyz"], data_dict["fp2_features"]
        data_dict["seed_inds"], data_dict["seed_xyz"], data_dict["seed_features"] = data_dict["fp2_inds"], xyz, features
        xyz, features = self.voting_net(xyz, features)
        features = features.div(torch.norm(features, p=2, dim=1).unsqueeze(1))
        data_dict["vote_xyz"], data_dict["vote_features"] = xyz, features
        return self.proposal_net(xyz, features, data_dict)

    # ---- the three stages of the path; forward() chains them, pipeline.PhasedTrainStep schedules them ----------
    def encode_image(self, data_dict):
        """ViT over view 0 of `images` (qa_module.py:611-620 -> blip_vqa_3d visual_encoder)"""
        return self.blip_model.visual_encoder(data_dict["images"][:, 0])

    def detect_objects(self, data_dict):
        """detector + the proposal-feature projection the fusion consumes (qa_module.py:438-479, 219-221)"""
        data_dict = self.detect(data_dict)
        feats = data_dict["aggregated_vote_features"]
        lin = self.object_feat_linear[0]
        from . import fusion_ops as ops
        if feats.is_cuda and ops.compute_dtype() == torch.bfloat16 and self.use_blip:
            # Linear + GELU as one launch of the MFMA GEMM family (bias + GELU epilogue): the fusion reads these tokens in
            # bf16 anyway; this was the detector stream's last library GEMM
            data_dict["object_feat"] = ops.linear(feats, lin.weight, lin.bias, act="gelu")
        else:
            data_dict["object_feat"] = self.object_feat_linear(feats)
        return data_dict

    def fuse(self, data_dict, image_embeds, object_feat=None, text_prep=None):
        """twin 2D/3D cross-attention encoder + answer decoder over the detector's proposals and the image tokens;
        text_prep: BLIP_VQA3D.prepare_text(question, answer) computed ahead of time (pipeline.PhasedTrainStep)"""
        if object_feat is None:
            object_feat = data_dict["object_feat"]
        object_mask = ~data_dict["bbox_mask"].bool().detach()  # True = not an object
        train = data_dict.get("phase", "train") == "train"
        out = self.blip_model(data_dict["images"][:, 0], data_dict["question"], image_embeds=image_embeds,
                              scene_object_embeds=object_feat.clone(), scene_object_mask=~object_mask,
                              answer=data_dict["answer"], train=train, k_test=256, data_dict=data_dict,
                              text_prep=text_prep if train else None)
        if train:
            data_dict["blip_loss"], data_dict["fused_feat"], data_dict["fused_mask"] = out
            data_dict["decoder_loss"] = data_dict["blip_loss"]  # the name lib/loss_helper.py reads (qa_module.py:696)
        else:
            data_dict["fused_feat"], data_dict["answer_scores"], data_dict["fused_mask"] = out
        if self.use_lang_cls or self.use_reference:
            qa_heads_forward(self, data_dict, object_feat, object_mask.unsqueeze(1).unsqueeze(2), data_dict["fused_feat"],
                             data_dict["fused_mask"], self.use_lang_cls, self.use_reference)
        return data_dict

    def forward(self, data_dict):
        """data_dict: point_clouds (B,N,3+C); with use_blip also images (B,V,3,H,W), question / answer
        (token dicts or strings).  Adds the detector outputs and, with BLIP, `blip_loss`, `fused_feat`."""
        runner = getattr(self, "_graphed", None)
        if runner is not None and runner.usable(data_dict):
            return runner.forward(data_dict)   # graphed.enable(model): HIP-graph replay behind the plain training loop
        image_embeds = None
        if self.use_blip and "images" in data_dict:
            from . import fusion_ops as ops
            if data_dict["images"].is_cuda and self.training:
                ops.new_step(data_dict["images"].device)  # fresh attention-dropout masks for this step
            image = data_dict["images"][:, 0]
            if ops.overlap_enabled(image):
                # image encoder || detector branch: FPS / ball query occupy B workgroups, the ViT wants the rest
                with ops.fork("image", image) as f:
                    f.uses(image)
                    image_embeds = self.encode_image(data_dict)
                data_dict = self.detect_objects(data_dict)
                f.join(image_embeds)
            else:
                data_dict = self.detect_objects(data_dict)
        else:
            data_dict = self.detect_objects(data_dict)
        if not self.use_blip:
            return data_dict
        return self.fuse(data_dict, image_embeds)
