"""Univl: the registry model `univl` (reference: prj/base_vtp/roi_univl/univl/model/univl_model.py:16-112).
Head types other than `video_text_retrieval` (MLM / ITM / classification / QA pre-training heads) are out of the
contrastive path's scope and raise."""
import torch

from antmmf.common.registry import registry
from antmmf.models.base_model import BaseModel
from .univl_video_ret import UnivlForVideoTextRetrieval


@registry.register_model("univl")
class Univl(BaseModel):
    def __init__(self, config):
        super().__init__(config)

    def build(self):
        if self.config.training_head_type != "video_text_retrieval":
            raise NotImplementedError(f"training_head_type {self.config.training_head_type!r} is outside the contrastive path")
        self.model = UnivlForVideoTextRetrieval(self.config)
        self.get_l2_input = self.model.module.get_l2_input

    def group_inputs(self, sample_list):
        groups = {"ocr": None, "caption": None, "region": None, "image": None, "generation": None}
        for key in sample_list.keys():
            for prefix in groups:
                if key.startswith(prefix):
                    if groups[prefix] is None:
                        groups[prefix] = {}
                    groups[prefix][key] = sample_list[key]
        return groups

    def forward(self, sample_list, *args, **kwargs):
        g = self.group_inputs(sample_list)
        out = self.model(g["image"], g["caption"], g["ocr"], g["region"], caption_output=g["generation"], sample_list=sample_list)
        return {"logits": out} if isinstance(out, torch.Tensor) else out

    def get_optimizer_parameters(self, config):
        return self.model.get_optimizer_parameters(config)
