"""Parameter layout of the RefTR model: names, shapes, storage order.

The NAMES are the reference's checkpoint contract (state_dict keys of ubc-vision/RefTR):
torchvision ResNet names under `img_backbone.0.body`, HuggingFace BertModel names under `lang_backbone`,
DETR-style names under `vl_transformer` (models/reftr.py:22-41, models/modeling/transformer.py:148-158,
208-219), plus the RefTR heads (models/reftr_transformer.py:82-125).  The ORDER is ours: all trainable
tensors live in one flat fp32 buffer, grouped by learning-rate group (main_vg.py:29-33,234-262) so the
fused AdamW kernel needs only three ranges, with BERT's query/key/value tensors adjacent so that the packed
[3H, H] projection is a free view of the master weights and of their gradients.
"""
from dataclasses import dataclass, field


@dataclass
class BertConfig:
    vocab_size: int = 30522
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    eps: float = 1e-12
    dropout: float = 0.1
    pad_idx: int = -1          # >= 0: RoBERTa-style position ids (HF create_position_ids_from_input_ids)


def roberta_config(layers=12):
    """roberta-base (configs/flickr30k/RefTR_flickr_roberta.sh:17): same encoder, vocabulary 50265, 514 positions offset
    by the padding index 1, one token type, LayerNorm eps 1e-5."""
    return BertConfig(vocab_size=50265, max_pos=514, type_vocab=1, eps=1e-5, pad_idx=1, layers=layers)


@dataclass
class ModelConfig:
    hidden: int = 256
    nheads: int = 8
    enc_layers: int = 6
    dec_layers: int = 6
    ffn: int = 2048
    dropout: float = 0.1
    max_lang_seq: int = 128
    n_q: int = 1                       # num_queries_per_phrase
    aux_loss: bool = True
    resnet_layers: tuple = (3, 4, 6, 3)   # resnet50; resnet101 = (3, 4, 23, 3)
    masks: bool = False                   # RefTRSeg: RES head (bbox_attention + mask_head), single phrase, no aux loss
    pos_learned: bool = False             # --position_embedding learned: PositionEmbeddingLearned (position_encoding.py:59-84)
    train_backbone: bool = True           # False: --lr_backbone 0 freezes the whole ResNet (backbone.py:87-89,150)
    dilation: bool = False                # --dilation (DC5): layer4 keeps stride 16, its 3x3 convolutions dilate by 2 (backbone.py:117-125)
    cem: bool = False                     # --ablation cem_loss: CEM block + loss_cem (reftr_segmentation.py:16-41, 62-64)
    bert: BertConfig = field(default_factory=BertConfig)


GROUP_MAIN, GROUP_BACKBONE, GROUP_BERT, GROUP_MASK = 0, 1, 2, 3


def lr_group(name):
    """main_vg.py:29-33: lr_backbone_names=['img_backbone.0'], lr_bert_names=['lang_backbone']."""
    if "img_backbone.0" in name:
        return GROUP_BACKBONE
    if "lang_backbone" in name:
        return GROUP_BERT
    if "bbox_attention" in name or "mask_head" in name:       # lr_mask_branch_names, main_vg.py:31,256-261
        return GROUP_MASK
    return GROUP_MAIN


def _bn(pfx, c, out):
    for k in ("weight", "bias", "running_mean", "running_var"):
        out.append((pfx + k, (c,), "buffer"))


def resnet_table(pfx, layers, train=True):
    """Returns [(name, shape, kind)], kind in {'param', 'frozen', 'buffer'} (backbone.py:87-89: conv1 and
    layer1 never train; FrozenBatchNorm2d tensors are buffers, backbone.py:52-57)."""
    t = [(pfx + "conv1.weight", (64, 3, 7, 7), "frozen")]
    _bn(pfx + "bn1.", 64, t)
    inpl = 64
    for li, n in enumerate(layers):
        planes = 64 * 2 ** li
        kind = "frozen" if (li == 0 or not train) else "param"
        for bi in range(n):
            p = f"{pfx}layer{li + 1}.{bi}."
            t.append((p + "conv1.weight", (planes, inpl, 1, 1), kind)); _bn(p + "bn1.", planes, t)
            t.append((p + "conv2.weight", (planes, planes, 3, 3), kind)); _bn(p + "bn2.", planes, t)
            t.append((p + "conv3.weight", (planes * 4, planes, 1, 1), kind)); _bn(p + "bn3.", planes * 4, t)
            if bi == 0:
                t.append((p + "downsample.0.weight", (planes * 4, inpl, 1, 1), kind)); _bn(p + "downsample.1.", planes * 4, t)
            inpl = planes * 4
    return t


def bert_table(pfx, bc: BertConfig):
    t = []
    e = pfx + "embeddings."
    t += [(e + "word_embeddings.weight", (bc.vocab_size, bc.hidden), "param"),
          (e + "position_embeddings.weight", (bc.max_pos, bc.hidden), "param"),
          (e + "token_type_embeddings.weight", (bc.type_vocab, bc.hidden), "param"),
          (e + "LayerNorm.weight", (bc.hidden,), "param"), (e + "LayerNorm.bias", (bc.hidden,), "param")]
    for i in range(bc.layers):
        lp = f"{pfx}encoder.layer.{i}."
        # q, k, v adjacent (weights then biases): packed [3H, H] / [3H] views
        for n in ("query", "key", "value"):
            t.append((lp + f"attention.self.{n}.weight", (bc.hidden, bc.hidden), "param"))
        for n in ("query", "key", "value"):
            t.append((lp + f"attention.self.{n}.bias", (bc.hidden,), "param"))
        t += [(lp + "attention.output.dense.weight", (bc.hidden, bc.hidden), "param"),
              (lp + "attention.output.dense.bias", (bc.hidden,), "param"),
              (lp + "attention.output.LayerNorm.weight", (bc.hidden,), "param"),
              (lp + "attention.output.LayerNorm.bias", (bc.hidden,), "param"),
              (lp + "intermediate.dense.weight", (bc.inter, bc.hidden), "param"),
              (lp + "intermediate.dense.bias", (bc.inter,), "param"),
              (lp + "output.dense.weight", (bc.hidden, bc.inter), "param"),
              (lp + "output.dense.bias", (bc.hidden,), "param"),
              (lp + "output.LayerNorm.weight", (bc.hidden,), "param"),
              (lp + "output.LayerNorm.bias", (bc.hidden,), "param")]
    t += [(pfx + "pooler.dense.weight", (bc.hidden, bc.hidden), "param"), (pfx + "pooler.dense.bias", (bc.hidden,), "param")]
    return t


def main_table(cfg: ModelConfig):
    E, F_ = cfg.hidden, cfg.ffn
    t = []

    def lin(p, o, i):
        t.append((p + "weight", (o, i), "param")); t.append((p + "bias", (o,), "param"))

    def ln(p, d):
        t.append((p + "weight", (d,), "param")); t.append((p + "bias", (d,), "param"))

    def mha(p):
        t.append((p + "in_proj_weight", (3 * E, E), "param")); t.append((p + "in_proj_bias", (3 * E,), "param"))
        lin(p + "out_proj.", E, E)

    def mlp_mapping(p, i, o):
        lin(p + "0.", o, i); ln(p + "1.", o); lin(p + "4.", o, o); ln(p + "5.", o)

    for i, (o, k) in enumerate(((E, E), (E, E), (4, E))):
        lin(f"bbox_embed.layers.{i}.", o, k)
    if cfg.pos_learned:                   # Joiner[1] = PositionEmbeddingLearned(E // 2): two nn.Embedding(50, E // 2), main lr group
        t.append(("img_backbone.1.row_embed.weight", (50, E // 2), "param"))
        t.append(("img_backbone.1.col_embed.weight", (50, E // 2), "param"))
    vt = "vl_transformer."
    if cfg.dec_layers > 0:
        ln(vt + "decoder.norm.", E)
    for i in reversed(range(cfg.dec_layers)):
        p = f"{vt}decoder.layers.{i}."
        mha(p + "self_attn."); mha(p + "multihead_attn.")
        lin(p + "linear1.", F_, E); lin(p + "linear2.", E, F_)
        ln(p + "norm1.", E); ln(p + "norm2.", E); ln(p + "norm3.", E)
    q = "query_encoder."
    t.append((q + "query_embed.weight", (cfg.n_q, 2 * E), "param"))
    lin(q + "linear1.", E, E); lin(q + "linear2.", E, E); lin(q + "linear3.", E, E)
    mlp_mapping(q + "fuse_encoder_query.", 2 * E, E)
    lin(q + "context_out.0.", E, E); ln(q + "context_out.1.", E)
    for i in reversed(range(cfg.enc_layers)):
        p = f"{vt}encoder.layers.{i}."
        mha(p + "self_attn."); lin(p + "linear1.", F_, E); lin(p + "linear2.", E, F_)
        ln(p + "norm1.", E); ln(p + "norm2.", E)
    t.append((vt + "level_embed", (1, E), "param"))
    t.append((vt + "lang_pos_embeddings.weight", (cfg.max_lang_seq, E), "param"))
    t.append((vt + "token_type_embeddings.weight", (2, E), "param"))
    mlp_mapping("map_sentence.", cfg.bert.hidden, E)
    mlp_mapping("map_phrase.", cfg.bert.hidden, E)
    t.append(("input_proj.0.0.weight", (E, 2048, 1, 1), "param")); t.append(("input_proj.0.0.bias", (E,), "param"))
    ln("input_proj.0.1.", E)
    return t


def pad64(c):
    return (c + 63) // 64 * 64


def seg_convs(cfg: ModelConfig):
    """MaskHeadSmallConv (models/reftr_segmentation.py:216-232): [(name, cin, cout, k)] in forward order."""
    E = cfg.hidden
    dim = 2 * E + cfg.nheads
    inter = [dim, E // 2, E // 4, E // 8, E // 16, E // 64]
    mh = "mask_head."
    convs = [(mh + "lay1", dim, dim, 3), (mh + "lay2", dim, inter[1], 3), (mh + "lay3", inter[1], inter[2], 3),
             (mh + "lay4", inter[2], inter[3], 3), (mh + "lay5", inter[3], inter[4], 3), (mh + "out_lay", inter[4], 1, 3),
             (mh + "adapter1", 1024, inter[1], 1), (mh + "adapter2", 512, inter[2], 1), (mh + "adapter3", 256, inter[3], 1)]
    return convs


def seg_table(cfg: ModelConfig):
    """bbox_attention + mask_head tensors (reference names, reftr_segmentation.py:59-60)."""
    E = cfg.hidden
    t = []
    for n in ("q_linear.", "k_linear."):
        t.append(("bbox_attention." + n + "weight", (E, E), "param")); t.append(("bbox_attention." + n + "bias", (E,), "param"))
    for name, ci, co, k in seg_convs(cfg):
        t.append((name + ".weight", (co, ci, k, k), "param")); t.append((name + ".bias", (co,), "param"))
    for i, (_, _, co, _) in enumerate(seg_convs(cfg)[:5]):
        t.append((f"mask_head.gn{i + 1}.weight", (co,), "param")); t.append((f"mask_head.gn{i + 1}.bias", (co,), "param"))
    return t


def cem_table(cfg: ModelConfig):
    """CEM block (reftr_segmentation.py:16-23): c1 Linear(E, 1), c2 Linear(E/16, 1), c3 Linear(E, E/16)."""
    E = cfg.hidden
    return [("cem_block.c1.weight", (1, E), "param"), ("cem_block.c1.bias", (1,), "param"),
            ("cem_block.c2.weight", (1, E // 16), "param"), ("cem_block.c2.bias", (1,), "param"),
            ("cem_block.c3.weight", (E // 16, E), "param"), ("cem_block.c3.bias", (E // 16,), "param")]


def phys_dims(cfg: ModelConfig):
    """Physical (stored) shapes of tensors whose channel counts are padded to multiples of 64 so that they are legal
    implicit-GEMM operands: conv weight [Cout_pad][kh][kw][Cin_pad], bias [Cout_pad]; the padding stays zero (zero
    gradients, zero AdamW updates).  The logical tensors are strided views of the unpadded corner."""
    d = {}
    if cfg.masks:
        for name, ci, co, k in seg_convs(cfg):
            d[name + ".weight"] = (pad64(co), k, k, pad64(ci))
            d[name + ".bias"] = (pad64(co),)
    if cfg.masks and cfg.cem:              # one-element biases: stored in 4-element (16-B) slots like every tensor of the flat buffers
        d["cem_block.c1.bias"] = (4,); d["cem_block.c2.bias"] = (4,)
    return d


def full_table(cfg: ModelConfig):
    """All tensors of the model.  Trainable ones are listed group by group (main, backbone, bert) in the
    order they are laid out in the flat parameter buffer."""
    return main_table(cfg) + (seg_table(cfg) if cfg.masks else []) + (cem_table(cfg) if cfg.masks and cfg.cem else []) + resnet_table("img_backbone.0.body.", cfg.resnet_layers, cfg.train_backbone) + bert_table("lang_backbone.", cfg.bert)


def reference_param_order(cfg: ModelConfig):
    """Names of the TRAINABLE tensors in the order the reference's `model.named_parameters()` yields them (module
    registration order of RefTR / RefTRSeg, torchvision ResNet, HF BertModel, DETR-style transformer layers).  A
    torch.optim.AdamW state_dict refers to parameters by their index in this order within each param group
    (main_vg.py:234-262), so this is what converts optimizer state to / from reference checkpoints."""
    E = cfg.hidden
    names = []
    pfx = "img_backbone.0.body."
    for li, n in enumerate(cfg.resnet_layers):
        if li == 0 or not cfg.train_backbone:
            continue                      # conv1 / layer1 (or, with lr_backbone 0, everything) are frozen (backbone.py:87-89): in no param group
        for bi in range(n):
            p = f"{pfx}layer{li + 1}.{bi}."
            for c in ("conv1", "conv2", "conv3"):
                names.append(p + c + ".weight")
            if bi == 0:
                names.append(p + "downsample.0.weight")
    if cfg.pos_learned:                   # img_backbone = Joiner(backbone, position_embedding): module 1 follows module 0
        names += ["img_backbone.1.row_embed.weight", "img_backbone.1.col_embed.weight"]
    lb = "lang_backbone."
    e = lb + "embeddings."
    names += [e + "word_embeddings.weight", e + "position_embeddings.weight", e + "token_type_embeddings.weight",
              e + "LayerNorm.weight", e + "LayerNorm.bias"]

    def wb(p):
        names.append(p + "weight"); names.append(p + "bias")
    for i in range(cfg.bert.layers):
        lp = f"{lb}encoder.layer.{i}."
        for n in ("attention.self.query.", "attention.self.key.", "attention.self.value.", "attention.output.dense.",
                  "attention.output.LayerNorm.", "intermediate.dense.", "output.dense.", "output.LayerNorm."):
            wb(lp + n)
    wb(lb + "pooler.dense.")
    vt = "vl_transformer."
    names += [vt + "level_embed", vt + "lang_pos_embeddings.weight", vt + "token_type_embeddings.weight"]

    def mha(p):
        names.append(p + "in_proj_weight"); names.append(p + "in_proj_bias"); wb(p + "out_proj.")
    for i in range(cfg.enc_layers):
        p = f"{vt}encoder.layers.{i}."
        mha(p + "self_attn."); wb(p + "linear1."); wb(p + "linear2."); wb(p + "norm1."); wb(p + "norm2.")
    for i in range(cfg.dec_layers):
        p = f"{vt}decoder.layers.{i}."
        mha(p + "self_attn."); mha(p + "multihead_attn."); wb(p + "linear1."); wb(p + "linear2.")
        wb(p + "norm1."); wb(p + "norm2."); wb(p + "norm3.")
    if cfg.dec_layers > 0:
        wb(vt + "decoder.norm.")
    for i in range(3):
        wb(f"bbox_embed.layers.{i}.")
    for m in ("map_sentence.", "map_phrase."):
        for j in ("0.", "1.", "4.", "5."):
            wb(m + j)
    q = "query_encoder."
    names.append(q + "query_embed.weight")
    for n in ("linear1.", "linear2.", "linear3.", "fuse_encoder_query.0.", "fuse_encoder_query.1.", "fuse_encoder_query.4.",
              "fuse_encoder_query.5.", "context_out.0.", "context_out.1."):
        wb(q + n)
    wb("input_proj.0.0."); wb("input_proj.0.1.")
    if cfg.masks:
        wb("bbox_attention.q_linear."); wb("bbox_attention.k_linear.")
        mh = "mask_head."
        for i in range(1, 6):
            wb(f"{mh}lay{i}."); wb(f"{mh}gn{i}.")
        wb(mh + "out_lay.")
        for i in range(1, 4):
            wb(f"{mh}adapter{i}.")
        if cfg.cem:                       # registered last (reftr_segmentation.py:62-64)
            wb("cem_block.c1."); wb("cem_block.c2."); wb("cem_block.c3.")
    return names
