"""state_dict key -> shape table of the RefTR model (REC head), built from the architecture alone.

TEST INFRASTRUCTURE (oracle/).  The key names are the reference's checkpoint contract:
torchvision names under img_backbone.0.body, HF names under lang_backbone, DETR names under
vl_transformer.  oracle/gen_golden.py asserts this table equals the imported reference model's state_dict.
"""


def resnet_shapes(pfx, layers=(3, 4, 6, 3)):
    s = {}

    def bn(p, c):
        for k in ("weight", "bias", "running_mean", "running_var"):
            s[p + k] = (c,)
    s[pfx + "conv1.weight"] = (64, 3, 7, 7)
    bn(pfx + "bn1.", 64)
    inpl = 64
    for li, n in enumerate(layers):
        planes = 64 * 2 ** li
        for bi in range(n):
            p = f"{pfx}layer{li + 1}.{bi}."
            s[p + "conv1.weight"] = (planes, inpl, 1, 1); bn(p + "bn1.", planes)
            s[p + "conv2.weight"] = (planes, planes, 3, 3); bn(p + "bn2.", planes)
            s[p + "conv3.weight"] = (planes * 4, planes, 1, 1); bn(p + "bn3.", planes * 4)
            if bi == 0:
                s[p + "downsample.0.weight"] = (planes * 4, inpl, 1, 1); bn(p + "downsample.1.", planes * 4)
            inpl = planes * 4
    return s


def bert_shapes(pfx, bc):
    s = {}

    def lin(p, o, i):
        s[p + "weight"] = (o, i); s[p + "bias"] = (o,)

    def ln(p, d):
        s[p + "weight"] = (d,); s[p + "bias"] = (d,)
    e = pfx + "embeddings."
    s[e + "word_embeddings.weight"] = (bc.vocab_size, bc.hidden)
    s[e + "position_embeddings.weight"] = (bc.max_pos, bc.hidden)
    s[e + "token_type_embeddings.weight"] = (bc.type_vocab, bc.hidden)
    ln(e + "LayerNorm.", bc.hidden)
    for i in range(bc.layers):
        lp = f"{pfx}encoder.layer.{i}."
        for n in ("query", "key", "value"):
            lin(lp + f"attention.self.{n}.", bc.hidden, bc.hidden)
        lin(lp + "attention.output.dense.", bc.hidden, bc.hidden); ln(lp + "attention.output.LayerNorm.", bc.hidden)
        lin(lp + "intermediate.dense.", bc.inter, bc.hidden)
        lin(lp + "output.dense.", bc.hidden, bc.inter); ln(lp + "output.LayerNorm.", bc.hidden)
    lin(pfx + "pooler.dense.", bc.hidden, bc.hidden)
    return s


def param_shapes(cfg):
    E, F_ = cfg.hidden, cfg.ffn
    s = {}

    def lin(p, o, i):
        s[p + "weight"] = (o, i); s[p + "bias"] = (o,)

    def ln(p, d):
        s[p + "weight"] = (d,); s[p + "bias"] = (d,)

    def mha(p):
        s[p + "in_proj_weight"] = (3 * E, E); s[p + "in_proj_bias"] = (3 * E,)
        lin(p + "out_proj.", E, E)

    def mlp_mapping(p, i, o):
        lin(p + "0.", o, i); ln(p + "1.", o); lin(p + "4.", o, o); ln(p + "5.", o)

    s.update(resnet_shapes("img_backbone.0.body.", cfg.resnet_layers))
    if getattr(cfg, "pos_learned", False):
        s["img_backbone.1.row_embed.weight"] = (50, E // 2); s["img_backbone.1.col_embed.weight"] = (50, E // 2)
    s.update(bert_shapes("lang_backbone.", cfg.bert))
    vt = "vl_transformer."
    s[vt + "level_embed"] = (1, E)
    s[vt + "lang_pos_embeddings.weight"] = (cfg.max_lang_seq, E)
    s[vt + "token_type_embeddings.weight"] = (2, E)
    for i in range(cfg.enc_layers):
        p = f"{vt}encoder.layers.{i}."
        mha(p + "self_attn."); lin(p + "linear1.", F_, E); lin(p + "linear2.", E, F_)
        ln(p + "norm1.", E); ln(p + "norm2.", E)
    for i in range(cfg.dec_layers):
        p = f"{vt}decoder.layers.{i}."
        mha(p + "self_attn."); mha(p + "multihead_attn.")
        lin(p + "linear1.", F_, E); lin(p + "linear2.", E, F_)
        ln(p + "norm1.", E); ln(p + "norm2.", E); ln(p + "norm3.", E)
    if cfg.dec_layers > 0:
        ln(vt + "decoder.norm.", E)
    for i, (o, k) in enumerate(((E, E), (E, E), (4, E))):
        lin(f"bbox_embed.layers.{i}.", o, k)
    mlp_mapping("map_sentence.", cfg.bert.hidden, E)
    mlp_mapping("map_phrase.", cfg.bert.hidden, E)
    q = "query_encoder."
    s[q + "query_embed.weight"] = (cfg.n_q, 2 * E)
    lin(q + "linear1.", E, E); lin(q + "linear2.", E, E); lin(q + "linear3.", E, E)
    mlp_mapping(q + "fuse_encoder_query.", 2 * E, E)
    lin(q + "context_out.0.", E, E); ln(q + "context_out.1.", E)
    s["input_proj.0.0.weight"] = (E, 2048, 1, 1); s["input_proj.0.0.bias"] = (E,)
    ln("input_proj.0.1.", E)
    if getattr(cfg, "masks", False):      # RefTRSeg: bbox_attention + mask_head (reftr_segmentation.py:59-60,180-238)
        lin("bbox_attention.q_linear.", E, E); lin("bbox_attention.k_linear.", E, E)
        dim = 2 * E + cfg.nheads
        inter = [dim, E // 2, E // 4, E // 8, E // 16, E // 64]
        mh = "mask_head."
        for i, (ci, co) in enumerate(((dim, dim), (dim, inter[1]), (inter[1], inter[2]), (inter[2], inter[3]), (inter[3], inter[4]))):
            s[f"{mh}lay{i + 1}.weight"] = (co, ci, 3, 3); s[f"{mh}lay{i + 1}.bias"] = (co,)
            ln(f"{mh}gn{i + 1}.", co)
        s[mh + "out_lay.weight"] = (1, inter[4], 3, 3); s[mh + "out_lay.bias"] = (1,)
        for i, (fd, co) in enumerate(((1024, inter[1]), (512, inter[2]), (256, inter[3]))):
            s[f"{mh}adapter{i + 1}.weight"] = (co, fd, 1, 1); s[f"{mh}adapter{i + 1}.bias"] = (co,)
        if getattr(cfg, "cem", False):    # CEM block (reftr_segmentation.py:16-23, 62-64)
            lin("cem_block.c1.", 1, E); lin("cem_block.c2.", 1, E // 16); lin("cem_block.c3.", E // 16, E)
    return s
