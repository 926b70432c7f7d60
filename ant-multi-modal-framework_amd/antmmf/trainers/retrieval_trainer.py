"""RetrievalTrainer: BaseTrainer + text<->video retrieval evaluation (reference: antmmf/trainers/retrieval_trainer.py:23-293).

Same registry name ("retrieval_trainer"), `evaluate_set` / `_evaluate_set` protocol and batch fields (`caption_raw_input_ids`,
`caption_input_mask`, `caption_tid`, `caption_vid_list`, `image_data`, `image_pad_mask`, `image_n_clips`, `image_num_frames`, `image_vid`,
`image_tid_list`): every caption batch and every DISTINCT video batch is encoded once, the model scores each (text batch, video batch)
block from the cached stage-1 outputs (`text_stage1_output` / `visual_stage1_output` in the sample list, univl_video_ret.py:466-476), blocks
are split over the ranks, and the metric object (GlobalRetrievalRecall) collects them with the ground-truth lists.

MI355X re-design of the same flow: the reference moves every block to the CPU (`nested_cpu`), pickles the Python result objects through
`all_gather` and calls `torch.cuda.empty_cache()` per block; here the similarity blocks stay on the device, each rank packs its blocks
into ONE flat fp32 buffer, a single `all_gather_into_tensor` exchanges them (block shapes are known to every rank from the cached
feature batches), and the ranks / recalls come from the rank-counting kernel."""
import torch

from antmmf.common.registry import registry
from antmmf.structures.sample import SampleList
from antmmf.trainers.base_trainer import BaseTrainer
from antmmf.utils.distributed_utils import get_rank, get_world_size, is_main_process, synchronize

SIMI_KEYS = ("l1_simi", "l2_simi", "l3_simi")


def split_batch(input_batch):
    visual_batch, text_batch = SampleList(), SampleList()
    for key in input_batch.keys():
        if "image" in key:
            visual_batch[key] = input_batch[key]
        if "caption" in key:
            text_batch[key] = input_batch[key]
    return visual_batch, text_batch


@registry.register_trainer("retrieval_trainer")
class RetrievalTrainer(BaseTrainer):
    def load(self):
        super().load()
        from antmmf.modules.metrics.global_retrieval_recall import GlobalRetrievalRecall

        keys = self.config.training_parameters.get("retrieval_simi_keys", None)
        self.overall_metric_evaluator = GlobalRetrievalRecall(simi_logit_key=list(keys) if keys else None) if keys else None
        self._metric_cls = GlobalRetrievalRecall

    def evaluate(self, batches):
        _, result = self.evaluate_set(batches)
        return result

    def evaluate_set(self, batches):
        self._enter_eval_rows()      # (the evaluation loader's own ragged-batch padding maximum: BaseTrainer.__init__)
        try:
            out = self._evaluate_set(batches)
        finally:
            self._leave_eval_rows()
        synchronize()
        return out

    def _retrieval_model(self):
        m = self.model.module if hasattr(self.model, "module") and hasattr(self.model.module, "model") else self.model
        return m, m.model.module  # (registry model, UnivlVideoBase)

    @torch.no_grad()
    def _evaluate_set(self, batches):
        reg_model, base = self._retrieval_model()
        was_training = self.model.training
        self.model.eval()
        dev = self.device
        # step 1: caption batches; video batches de-duplicated by video id (a video with several captions appears once)
        seen, text_batches, visual_batches, dataset_name = set(), [], [], None
        for batch in batches:
            batch = batch.to(dev) if isinstance(batch, SampleList) else SampleList(batch).to(dev)
            dataset_name = batch.get("dataset_name", dataset_name)
            vb, tb = split_batch(batch)
            text_batches.append((tb["caption_raw_input_ids"], tb["caption_input_mask"], tb["caption_tid"], tb["caption_vid_list"]))
            vids = [int(v) for v in vb["image_vid"].tolist()]
            keep = [i for i, v in enumerate(vids) if not (v in seen or seen.add(v))]
            if not keep:
                continue
            idx = torch.tensor(keep, device=dev)
            visual_batches.append((vb["image_data"].index_select(0, idx), vb["image_pad_mask"].index_select(0, idx),
                                   [vb["image_n_clips"][i] for i in keep], [vb["image_num_frames"][i] for i in keep],
                                   vb["image_vid"].index_select(0, idx), [vb["image_tid_list"][i] for i in keep]))
        # step 2: stage-1 features, once per batch, sorted by id as the reference does
        cross = base.with_cross_encoder or getattr(base, "need_cross_inputs", False)  # stage-2 / stage-3 heads want the cross-embedded tokens
        text_feats = []
        for ids, mask, tid, vid_list in text_batches:
            pooled = base.forward_text_encoder(ids, mask)["pooled_output"]
            if cross:
                cap_embed, cap_mask, bsz = base.prepare_cross_text(ids, mask)
            else:
                cap_embed, cap_mask, bsz = None, mask, pooled.shape[0]
            text_feats.append(((cap_embed, cap_mask, pooled, bsz, None), int(tid.min()), vid_list))
        text_feats.sort(key=lambda x: x[1])
        visual_feats = []
        for data, mask, n_clips, n_frames, vids, tid_list in visual_batches:
            vd = base.forward_img_encoder(data, mask, n_clips, n_frames)
            if cross:
                vis_embed, vis_mask, num_clip = base.prepare_cross_visual(vd["visual_embed"], vd["visual_mask"])
            else:
                vis_embed, vis_mask, num_clip = vd["visual_embed"], vd["visual_mask"], vd["visual_embed"].shape[1]
            visual_feats.append(((vis_embed, vis_mask, vd["clip_feature"], num_clip), int(vids.min()), tid_list))
        visual_feats.sort(key=lambda x: x[1])
        text2video = [x[2] for x in text_feats]
        video2text = [x[2] for x in visual_feats]
        # step 3: blocks (idx_t, idx_v) dealt round-robin over the ranks; every block's similarity matrices stay on the device
        pairs = [(t, v) for t in range(len(text_feats)) for v in range(len(visual_feats))]
        world, rank = get_world_size(), get_rank()
        mine, keys = {}, None
        for p, (t, v) in enumerate(pairs):
            if p % world != rank:
                continue
            sl = SampleList(text_stage1_output=text_feats[t][0], visual_stage1_output=visual_feats[v][0], dataset_type="val",
                            dataset_name=dataset_name)
            out = reg_model(sl)
            keys = [k for k in SIMI_KEYS if k in out]
            mine[(t, v)] = {k: out[k].float() for k in keys}
        if keys is None:  # a rank without blocks: the key set is that of the training stages
            stages = str(reg_model.config.training_stage)
            keys = [k for k, s in zip(SIMI_KEYS, ("stage1", "stage2", "stage3")) if s in stages]
        # step 4: one all-gather of every rank's packed blocks
        shapes = {(t, v): (text_feats[t][0][3], len(visual_feats[v][2])) for (t, v) in pairs}
        blocks = self._exchange(mine, pairs, shapes, keys, world, rank, dev)
        # step 5: metrics in (idx_t, idx_v) order
        metric = self.overall_metric_evaluator or self._metric_cls(simi_logit_key=keys)
        metric.reset() if hasattr(metric, "reset") else None
        for (t, v) in pairs:
            metric.collect(None, blocks[(t, v)], t, v, t2v=text2video[t], v2t=video2text[v])
        result = {k: float(v) for k, v in metric.summarize().items()}
        if was_training:
            self.model.train()
        return dataset_name, result

    @staticmethod
    def _exchange(mine, pairs, shapes, keys, world, rank, dev):
        if world == 1:
            return mine
        import torch.distributed as dist

        per_rank = [[pr for p, pr in enumerate(pairs) if p % world == r] for r in range(world)]
        size = lambda pr: len(keys) * shapes[pr][0] * shapes[pr][1]  # noqa: E731
        longest = max(sum(size(pr) for pr in prs) for prs in per_rank)
        buf = torch.zeros(max(longest, 1), dtype=torch.float32, device=dev)
        off = 0
        for pr in per_rank[rank]:
            for k in keys:
                n = shapes[pr][0] * shapes[pr][1]
                buf[off:off + n] = mine[pr][k].reshape(-1)
                off += n
        every = torch.empty(world * buf.numel(), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(every, buf)
        blocks = {}
        for r in range(world):
            off = r * buf.numel()
            for pr in per_rank[r]:
                blocks[pr] = {}
                for k in keys:
                    n = shapes[pr][0] * shapes[pr][1]
                    blocks[pr][k] = every[off:off + n].view(shapes[pr])
                    off += n
        return blocks
