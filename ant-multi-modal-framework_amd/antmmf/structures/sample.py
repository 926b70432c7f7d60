"""SampleList: the batch dict the trainer hands to the model (reference: antmmf/structures/sample.py:58-334):
an OrderedDict with attribute access and `.to(device)`; keys are routed to the model by prefix
(`image_*`, `caption_*`; prj/base_vtp/roi_univl/univl/model/univl_model.py:36-51)."""
import collections

import torch


class SampleList(collections.OrderedDict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to(self, device, non_blocking=True):
        out = SampleList()
        for k, v in self.items():
            out[k] = v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v
        return out

    def get_batch_size(self):
        for v in self.values():
            if torch.is_tensor(v):
                return v.shape[0]
        return 0
