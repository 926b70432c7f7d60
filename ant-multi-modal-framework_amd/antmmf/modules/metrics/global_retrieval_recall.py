"""GlobalRetrievalRecall on the MI355X path (reference: antmmf/modules/metrics/global_retrieval_recall.py:13-195).

Same registry name, constructor kwargs (`simi_logit_key`), collect / calculate / summarize protocol and result keys
("<key>_r@1", "<key>_t2v-mr", ...).  The reference moves every similarity block to numpy, argsorts the full matrix on one CPU
core and walks the rows in Python; here blocks stay on the GPU and the rank of each row's ground truth comes from one fused
counting kernel (`antmmf_rank_rows`: rank = position in a stable descending sort = strictly larger scores + equal scores at a lower
column index), after which recall@k / median rank are reductions over a [rows] vector.  Ties are broken by column index (the
reference's position inside a tie group depends on numpy's sort implementation); a NaN ground-truth score ranks last, so a collapsed
or diverged model cannot report perfect retrieval."""
from collections import defaultdict

import torch

from antmmf.common.registry import registry
from antmmf.hip import ops
from antmmf.modules.metrics.base_metric import BaseMetric


def _csr(gt_lists, device):
    off, idx = [0], []
    for g in gt_lists:
        g = sorted(set(int(x) for x in g))
        idx.extend(g)
        off.append(len(idx))
    return (torch.tensor(off, dtype=torch.int32, device=device), torch.tensor(idx, dtype=torch.int32, device=device))


def gt_ranks(sim_matrix, gt_lists=None):
    """rank (0 = first) of the best ground truth of every row; gt_lists=None means the diagonal."""
    S = sim_matrix.float().contiguous()
    if gt_lists is None:
        gt_lists = [[i] for i in range(S.shape[0])]
    off, idx = _csr(gt_lists, S.device)
    return ops.rank_rows(S, off, idx)


def _recall_dict(rank, prefix=""):
    r = rank.float()
    out = {prefix + f"r@{k}": float((r < k).float().mean()) for k in (1, 5, 10)}
    out[prefix + "mr"] = float(_np_median(r)) + 1
    return out


def _np_median(r):
    """numpy's median (mean of the two middle values for even counts) on a device vector."""
    v = r.sort().values
    n = v.numel()
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def _cal_recall(sim_matrix):
    """Batch-wise metrics of a square matrix whose ground truth is the diagonal (reference :91-103)."""
    d = _recall_dict(gt_ranks(sim_matrix))
    return {"mr": d["mr"], "r@1": d["r@1"], "r@5": d["r@5"], "r@10": d["r@10"]}


def _cal_sym_recall(sim_matrix, t2v, v2t):
    """Text->video and video->text metrics with explicit ground-truth lists (reference :31-89)."""
    out = {}
    for prefix, S, gt in (("t2v-", sim_matrix, t2v), ("v2t-", sim_matrix.t(), v2t)):
        d = _recall_dict(gt_ranks(S, gt), prefix)
        out[prefix + "mean_recall"] = (d[prefix + "r@1"] + d[prefix + "r@5"] + d[prefix + "r@10"]) / 3.0
        out.update(d)
    return out


@registry.register_metric("global_retrieval_recall")
class GlobalRetrievalRecall(BaseMetric):
    def __init__(self, *args, **kwargs):
        super().__init__(name=kwargs.get("name", "global_retrieval_recall"))
        self._simi_logit_key = kwargs.get("simi_logit_key")
        self._ind = dict([(k, None) for k in self._simi_logit_key])
        self.gt_t2v = dict()
        self.gt_v2t = dict()

    def reset(self):
        for simi_level in self._simi_logit_key:
            self._ind[simi_level] = None
        # the ground-truth lists belong to the set being evaluated (the reference keeps them, global_retrieval_recall.py:120-122, so a second,
        # different set -- val then test -- would be scored against the first set's lists)
        self.gt_t2v, self.gt_v2t = dict(), dict()

    def collect(self, sample_list, model_output, idx_t, idx_v, t2v=None, v2t=None, **kwargs):
        if t2v is not None and idx_t not in self.gt_t2v:
            self.gt_t2v[idx_t] = t2v
        if v2t is not None and idx_v not in self.gt_v2t:
            self.gt_v2t[idx_v] = v2t
        for simi_level in self._simi_logit_key:
            if self._ind[simi_level] is None:
                self._ind[simi_level] = defaultdict(list)
            if simi_level in model_output:
                self._ind[simi_level][idx_t].append(model_output[simi_level].detach().float())  # stays on the device

    def calculate(self, sample_list, model_output, *args, **kwargs):
        score_dict = dict()
        for logit_key in self._simi_logit_key:
            if logit_key not in model_output:
                continue
            simi_matrix = model_output[logit_key]
            if simi_matrix.size(0) == simi_matrix.size(1):
                metric_dict = _cal_recall(simi_matrix)
            else:
                metric_dict = {"mr": 0.0, "r@1": 0.0, "r@5": 0.0, "r@10": 0.0}
            for name, val in metric_dict.items():
                score_dict["{}_{}".format(logit_key, name)] = torch.tensor(val, dtype=torch.float64)
        return score_dict

    def summarize(self, *args, **kwargs):
        score_dict = dict()
        t2v = [a for x in sorted(self.gt_t2v.items(), key=lambda x: x[0]) for a in x[1]]
        v2t = [a for x in sorted(self.gt_v2t.items(), key=lambda x: x[0]) for a in x[1]]
        for logit_key, logit_dict in self._ind.items():
            if not logit_dict:
                continue
            simi_matrix = torch.cat([torch.cat(v, 1) for k, v in sorted(logit_dict.items(), key=lambda x: x[0])], 0)
            for name, val in _cal_sym_recall(simi_matrix, t2v, v2t).items():
                score_dict["{}_{}".format(logit_key, name)] = torch.tensor(val, dtype=torch.float64)
        return score_dict
