"""BEiT3 dual-use backbone, vision-only or text-only input (reference: prj/M2_Encoder/vlmo/torchscale/model/BEiT3.py:16-96)."""
from torch import nn

from antmmf.hip import functional as HF
from ..architecture.encoder import Encoder
from ..component.embedding import PositionalEmbedding, TextEmbedding, VisionEmbedding
from ..component.multiway_network import MutliwayEmbedding


class BEiT3(nn.Module):
    def __init__(self, args, **kwargs):
        super().__init__()
        self.args = args
        assert args.multiway and args.vocab_size > 0
        self.text_embed = TextEmbedding(args.vocab_size, args.encoder_embed_dim)
        self.vision_embed = VisionEmbedding(args.img_size, args.patch_size, args.in_chans, args.encoder_embed_dim,
                                            contain_mask_token=True, prepend_cls_token=True)
        embed_positions = MutliwayEmbedding(modules=[
            PositionalEmbedding(self.vision_embed.num_position_embeddings() + 2, args.encoder_embed_dim),
            PositionalEmbedding(args.max_source_positions, args.encoder_embed_dim)], dim=1)
        self.encoder = Encoder(args, embed_tokens=None, embed_positions=embed_positions, output_projection=None)

    def forward(self, textual_tokens=None, visual_tokens=None, text_padding_position=None, attn_mask=None,
                vision_masked_position=None, incremental_state=None, positions=None, image_shift=0.0, image_scale=1.0):
        assert (textual_tokens is None) != (visual_tokens is None), "ITC towers take one modality at a time"
        assert attn_mask is None and vision_masked_position is None and incremental_state is None and positions is None
        if textual_tokens is None:
            ve = self.vision_embed
            n = ve.num_position_embeddings()
            pos = self.encoder.embed_positions.A.weight[2:n + 2]
            x = HF.patch_embed(visual_tokens, ve.proj.weight, ve.proj.bias, ve.cls_token, pos, ve.patch_size[0], image_shift, image_scale)
            return self.encoder(token_embeddings=x, encoder_padding_mask=None, multiway_split_position=-1, pos_added=True)
        pad = None
        zero_rows = None
        if text_padding_position is not None:
            pad = text_padding_position.bool()
            zero_rows = pad.to(dtype=__import__("torch").uint8).contiguous()
        x = HF.embed(textual_tokens, self.text_embed.weight, self.encoder.embed_positions.B.weight, None, zero_rows, pos_offset=2)
        return self.encoder(token_embeddings=x, encoder_padding_mask=pad, multiway_split_position=0, pos_added=True)
