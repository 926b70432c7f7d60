"""EncoderConfig: attribute bag with the field names / defaults the M2 towers read (reference:
prj/M2_Encoder/vlmo/torchscale/architecture/config.py:5-66).  MoE, relative-position and xpos options exist in
the reference but are disabled in every M2 config (moe_freq 0, rel_pos_buckets 0, xpos_rel_pos False); asking for
them raises."""


class EncoderConfig:
    def __init__(self, **kw):
        g = kw.pop
        self.encoder_embed_dim = g("encoder_embed_dim", 768)
        self.encoder_attention_heads = g("encoder_attention_heads", 12)
        self.encoder_ffn_embed_dim = g("encoder_ffn_embed_dim", 3072)
        self.encoder_layers = g("encoder_layers", 12)
        self.encoder_normalize_before = g("encoder_normalize_before", True)
        self.normalize_output = g("normalize_output", True)
        self.activation_fn = g("activation_fn", "gelu")
        self.dropout = g("dropout", 0.0)
        self.drop_path_rate = g("drop_path_rate", 0.0)
        self.attention_dropout = g("attention_dropout", 0.0)
        self.activation_dropout = g("activation_dropout", 0.0)
        self.no_scale_embedding = g("no_scale_embedding", True)
        self.layernorm_embedding = g("layernorm_embedding", False)
        self.moe_freq = g("moe_freq", 0)
        self.rel_pos_buckets = g("rel_pos_buckets", 0)
        self.max_rel_pos = g("max_rel_pos", 0)
        self.deepnorm = g("deepnorm", False)
        self.subln = g("subln", True)
        self.bert_init = g("bert_init", False)
        self.multiway = g("multiway", False)
        self.share_encoder_input_output_embed = g("share_encoder_input_output_embed", False)
        self.max_source_positions = g("max_source_positions", 1024)
        self.no_output_layer = g("no_output_layer", False)
        self.layernorm_eps = g("layernorm_eps", 1e-5)
        self.share_layer = g("share_layer", False)
        self.share_attn = g("share_attn", False)
        self.mask_ratio = g("mask_ratio", 0)
        self.max_text_len = g("max_text_len", 52)
        self.one_attn = g("one_attn", False)
        self.vocab_size = g("vocab_size", -1)
        self.img_size = g("img_size", 224)
        self.patch_size = g("patch_size", 16)
        self.in_chans = g("in_chans", 3)
        self.checkpoint_activations = g("checkpoint_activations", False)
        self.xpos_rel_pos = g("xpos_rel_pos", False)
        if self.deepnorm:
            self.encoder_normalize_before = False
            self.subln = False
        if self.subln:
            self.encoder_normalize_before = True
            self.deepnorm = False
        unsupported = dict(moe_freq=0, rel_pos_buckets=0, deepnorm=False, subln=True, share_layer=False, share_attn=False,
                           mask_ratio=0, one_attn=False, xpos_rel_pos=False, layernorm_embedding=False, activation_fn="gelu",
                           no_scale_embedding=True, encoder_normalize_before=True, normalize_output=True)
        for k, v in unsupported.items():
            if getattr(self, k) != v:
                raise NotImplementedError(f"EncoderConfig.{k}={getattr(self, k)!r}: only the M2 ITC setting ({v!r}) is on the HIP path")
        if max(self.dropout, self.attention_dropout, self.activation_dropout, self.drop_path_rate) > 0:
            raise NotImplementedError("M2 configs use dropout 0 everywhere; p > 0 is not implemented on the HIP path")
