"""BLIP image-text matching model and the question -> best-views ranking built on it -- mirror of the reference's
models/blip_itm.py (BLIP_ITM :10-70, blip_itm :73-79) and of the scoring in eval_scene_best_views.py:199-303 that writes
the `--i2tfile` ({"view", "answer", "itm_scores"}) utils/blip_utils.py:233-240 reads (offline).

Same class / constructor / forward signature / state-dict keys (visual_encoder.*, text_encoder.*, vision_proj, text_proj,
itm_head), so the BLIP retrieval checkpoints load.  The encoders are this package's `vit.VisionTransformer` and
`med.BertModel` (HIP attention / GEMM / LayerNorm kernels under `fusion_ops.set_compute_dtype(torch.bfloat16)`, torch
composition in fp32).

`encode_views` / `rank_views` restate the script's loop without its host round trips: the reference keeps every view's
full token sequence (577 x 1024 floats) on the CPU and re-uploads it per scene, although only the class token is used; here
the projected, normalised class-token feature (256 floats) of every view stays on the device, a scene's questions are
scored against its views with one matrix product, and the ranking leaves the device once.
"""
import torch
import torch.nn.functional as F
from torch import nn

from .blip_vqa_3d import DEFAULT_BLIP_CONFIG, _tokens, init_tokenizer, load_checkpoint
from .med import BertConfig, BertModel
from .vit import create_vit


class BLIP_ITM(nn.Module):
    def __init__(self, med_config=DEFAULT_BLIP_CONFIG, image_size=384, vit="base", vit_grad_ckpt=False, vit_ckpt_layer=0,
                 embed_dim=256):
        super().__init__()
        self.visual_encoder, vision_width = create_vit(vit, image_size, vit_grad_ckpt, vit_ckpt_layer)
        self.tokenizer = init_tokenizer()
        med_config = BertConfig.from_json_file(med_config) if isinstance(med_config, str) else med_config
        med_config.encoder_width = vision_width
        self.text_encoder = BertModel(config=med_config, add_pooling_layer=False)
        text_width = self.text_encoder.config.hidden_size
        self.vision_proj = nn.Linear(vision_width, embed_dim)
        self.text_proj = nn.Linear(text_width, embed_dim)
        self.itm_head = nn.Linear(text_width, 2)

    def forward(self, image, caption, match_head="itm"):
        image_embeds = self.visual_encoder(image)
        image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long, device=image.device)
        text = _tokens(caption, self.tokenizer, image.device, padding="max_length", truncation=True, max_length=80)
        if match_head == "itm":
            output = self.text_encoder(text.input_ids, attention_mask=text.attention_mask,
                                       encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                                       return_dict=True)
            return self.itm_head(output.last_hidden_state[:, 0, :].float())
        if match_head == "itc":
            return self.image_features(image_embeds) @ self.text_features(text).t()
        raise ValueError("match_head must be 'itm' or 'itc'")

    # ---- the two halves of the contrastive score (eval_scene_best_views.py:246-281) -----------------------------------------
    def image_features(self, image_embeds):
        return F.normalize(self.vision_proj(image_embeds[:, 0, :].float()), dim=-1)

    def text_features(self, text):
        out = self.text_encoder(text.input_ids, attention_mask=text.attention_mask, return_dict=True, mode="text")
        return F.normalize(self.text_proj(out.last_hidden_state[:, 0, :].float()), dim=-1)


def blip_itm(pretrained="", **kwargs):
    model = BLIP_ITM(**kwargs)
    if pretrained:
        model, msg = load_checkpoint(model, pretrained)
        print(msg)
    return model


@torch.no_grad()
def encode_views(model, images, batch_size=256):
    """images (N, 3, S, S) of one scene's views (any device) -> (N, embed_dim) normalised view features on the model's
    device (eval_scene_best_views.py:199-214 + :246-249, without keeping the token sequences)"""
    dev = model.vision_proj.weight.device
    feats = []
    for i in range(0, images.shape[0], batch_size):
        feats.append(model.image_features(model.visual_encoder(images[i:i + batch_size].to(dev, non_blocking=True))))
    return torch.cat(feats, 0)


@torch.no_grad()
def rank_views(model, view_feats, image_names, questions, question_ids, max_length=70):
    """eval_scene_best_views.py:255-287 for one scene: questions = list of strings, or {"input_ids", "attention_mask"}
    tensors -> (view, itm_scores): {question_id: image names, best first}, {question_id: their similarities (floats)}"""
    dev = view_feats.device
    text = _tokens(questions, model.tokenizer, dev, padding="max_length", truncation=True, max_length=max_length)
    sim = model.text_features(text) @ view_feats.t()                    # (questions, views)
    order = sim.argsort(dim=1, descending=True, stable=True)
    order_h, sim_h = order.cpu().tolist(), torch.gather(sim, 1, order).cpu().tolist()   # the one hand-over
    view = {str(q): [image_names[i] for i in order_h[k]] for k, q in enumerate(question_ids)}
    scores = {str(q): sim_h[k] for k, q in enumerate(question_ids)}
    return view, scores


def save_view_map(path, view, itm_scores, answer=None):
    """the `--outfile` / `--i2tfile` of the reference (eval_scene_best_views.py:301-303; read back by
    utils/blip_utils.py:233-240 with torch.load)"""
    torch.save({"view": view, "answer": answer or {}, "itm_scores": itm_scores}, path)
