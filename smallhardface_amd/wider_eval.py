"""WIDER FACE evaluation (easy / medium / hard AP) of written detections.

What the reference runs after ``test_net`` for the WIDER imdb (lib/datasets/wider.py:169-195 ->
lib/wider_eval_tools/wider_eval.py:10-222, itself a port of the official MATLAB toolbox).  Semantics kept exactly,
because they decide the published AP numbers:

  * predictions are read back from the per-image text files (x, y, w, h, score), ordered by falling score, and the
    scores are min-max normalised over the WHOLE prediction set (wider_eval.py:42-59);
  * boxes become (x1, y1, x1 + w, y1 + h); IoU uses the +1 pixel convention with a zero-union guard (:62-79);
  * ``mimic_eval_bug`` (cfg.MISC.MIMIC_EVAL_BUG, default on): every IoU is ROUNDED to 0 / 1 before the arg-max, so
    a detection is attributed to the FIRST ground-truth box with IoU >= 0.5, not the best one (:92-95);
  * a detection matched to a ground-truth box outside the setting's subset is neither a hit nor a false positive
    (``proposal_list = -1``), a box of the subset counts once (:96-103);
  * 1000 thresholds ``1 - (t + 1) / 1000`` on the normalised score (:106-120), dataset curve = summed counts
    (:123-130), AP = VOC-style area under the monotone precision envelope (:133-139).

The implementation is organised around in-memory structures (``WiderGT`` / lists of (n, 5) arrays) with separate
readers for the toolbox's ``.mat`` ground truth and the detection text files, and the per-image threshold sweep is a
``searchsorted`` instead of a 1000-step loop.  Host-side utility: nothing here touches the GPU.
"""
import os

import numpy as np

THRESH_NUM = 1000
SETTINGS = ('easy_val', 'medium_val', 'hard_val')


class WiderGT(object):
    """Ground truth of one setting, flattened over events: per image its name, event, (n, 4) x-y-w-h boxes and the
    0-based indices of the boxes that count in this setting."""

    def __init__(self, events, names, boxes, keep):
        self.events, self.names, self.boxes, self.keep = list(events), list(names), list(boxes), list(keep)

    def __len__(self):
        return len(self.names)


def load_gt_mat(path):
    """``wider_{easy,medium,hard}_val.mat`` / ``wider_face_val.mat`` of the official toolbox -> WiderGT (image order =
    event-major, the order the reference flattens with np.vstack / reduce, wider_eval.py:150-154)."""
    from scipy import io as sio
    m = sio.loadmat(path)
    events, names, boxes, keep = [], [], [], []
    for e in range(m['event_list'].shape[0]):
        ev = str(m['event_list'][e][0][0])
        files = m['file_list'][e][0]
        bbx = m['face_bbx_list'][e][0]
        sub = m['gt_list'][e][0] if 'gt_list' in m else None
        for j in range(files.shape[0]):
            events.append(ev)
            names.append(str(files[j][0][0]))
            b = np.asarray(bbx[j][0], dtype=np.float64).reshape(-1, 4)
            boxes.append(b)
            if sub is None:
                keep.append(np.arange(b.shape[0]))
            else:
                keep.append(np.asarray(sub[j][0], dtype=np.int64).reshape(-1) - 1)  # MATLAB indices are 1-based
    return WiderGT(events, names, boxes, keep)


def read_predictions(pred_dir, gt):
    """Per image of ``gt``: the (n, 5) x-y-w-h-score rows of ``<pred_dir>/<event>/<name>.txt`` (the files
    datasets.write_detections_wider produces), ordered by falling score; None when the file is missing or broken
    (the reference logs and carries on, wider_eval.py:33-37)."""
    out = []
    for ev, name in zip(gt.events, gt.names):
        try:
            with open(os.path.join(pred_dir, ev, name + '.txt')) as f:
                lines = [l.strip() for l in f.readlines()]
            n = int(lines[1])
            rows = np.array([[float(v) for v in lines[2 + k].split()] for k in range(n)], dtype=np.float64).reshape(n, 5)
            out.append(sort_by_score(rows))
        except Exception:
            out.append(None)
    return out


def sort_by_score(rows):
    """Falling score; equal scores keep their file order (the reference's ``argsort()[::-1]`` leaves ties undefined)."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 5)
    return rows[np.argsort(-rows[:, 4], kind='stable')]


def normalise_scores(preds):
    """Min-max over every prediction of the set (wider_eval.py:42-59).  Returns new arrays."""
    allp = [p[:, 4] for p in preds if p is not None and len(p)]
    lo = min(float(s.min()) for s in allp)
    hi = max(max(float(s.max()) for s in allp), 0.0)   # the reference starts its running maximum at 0
    out = []
    for p in preds:
        if p is None:
            out.append(None)
            continue
        q = np.array(p, dtype=np.float64)
        q[:, 4] = (q[:, 4] - lo) / (hi - lo)
        out.append(q)
    return out


def _overlaps(gt_xyxy, box):
    iw = np.minimum(gt_xyxy[:, 2], box[2]) - np.maximum(gt_xyxy[:, 0], box[0]) + 1
    ih = np.minimum(gt_xyxy[:, 3], box[3]) - np.maximum(gt_xyxy[:, 1], box[1]) + 1
    inter = iw * ih
    union = (gt_xyxy[:, 2] - gt_xyxy[:, 0] + 1) * (gt_xyxy[:, 3] - gt_xyxy[:, 1] + 1) + \
        (box[2] - box[0] + 1) * (box[3] - box[1] + 1) - inter
    union = np.where(union == 0, np.inf, union)
    o = inter / union
    o[(iw <= 0) | (ih <= 0)] = 0
    return o


def image_counts(pred, gt_boxes, keep, iou_thresh=0.5, mimic_eval_bug=True):
    """One image: (hits[h], is_proposal[h]) after each of the score-ordered detections -- the number of distinct
    subset faces found so far, and whether detection h counts as a proposal (False when it sits on a face outside
    the subset)."""
    n, g = pred.shape[0], gt_boxes.shape[0]
    counted = np.zeros(g, dtype=bool)
    counted[keep] = True
    gxy = np.array(gt_boxes, dtype=np.float64)
    gxy[:, 2:4] += gxy[:, 0:2]
    pxy = np.array(pred[:, :4], dtype=np.float64)
    pxy[:, 2:4] += pxy[:, 0:2]
    state = np.zeros(g, dtype=np.int64)      # 0 untouched, 1 found, -1 hit although outside the subset
    hits = np.zeros(n, dtype=np.int64)
    proposal = np.ones(n, dtype=bool)
    for h in range(n):
        o = _overlaps(gxy, pxy[h])
        if mimic_eval_bug:
            o = np.floor(o + 0.5)             # Python 2 round(): halves away from zero
        idx = int(np.argmax(o))
        if o[idx] >= iou_thresh:
            if not counted[idx]:
                state[idx] = -1
                proposal[h] = False
            elif state[idx] == 0:
                state[idx] = 1
        hits[h] = int((state == 1).sum())
    return hits, proposal


def image_pr_info(pred, hits, proposal, thresh_num=THRESH_NUM):
    """(thresh_num, 2): [number of proposals, number of subset faces found] with score >= each threshold."""
    t = np.arange(thresh_num, dtype=np.float64)
    thresh = 1 - (t + 1.) / thresh_num
    # pred is score-descending: the last row with score >= thresh is (count of such rows) - 1
    cnt = np.searchsorted(-pred[:, 4], -thresh, side='right')
    cum_prop = np.concatenate([[0], np.cumsum(proposal.astype(np.int64))])
    info = np.zeros((thresh_num, 2))
    has = cnt > 0
    info[has, 0] = cum_prop[cnt[has]]
    info[has, 1] = hits[cnt[has] - 1]
    return info


def voc_ap(rec, prec):
    mrec = np.hstack([0, rec, 1])
    mpre = np.hstack([0, prec, 0])
    for i in range(mpre.shape[0] - 2, -1, -1):
        mpre[i] = max(mpre[i], mpre[i + 1])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def evaluate_setting(norm_preds, gt, iou_thresh=0.5, mimic_eval_bug=True, thresh_num=THRESH_NUM):
    """PR curve (thresh_num, 2) = [precision, recall] of one setting (wider_eval.py:142-177)."""
    total = np.zeros((thresh_num, 2))
    count_face = 0
    for j in range(len(gt)):
        keep = np.asarray(gt.keep[j], dtype=np.int64).reshape(-1)
        count_face += keep.shape[0]
        p = norm_preds[j]
        if gt.boxes[j].size == 0 or p is None or p.size == 0:
            continue
        hits, proposal = image_counts(p, gt.boxes[j], keep, iou_thresh, mimic_eval_bug)
        total += image_pr_info(p, hits, proposal, thresh_num)
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.stack([total[:, 1] / total[:, 0], total[:, 1] / count_face], axis=1)


def evaluate(preds, gts, iou_thresh=0.5, mimic_eval_bug=True):
    """``preds``: per image (n, 5) x-y-w-h-score arrays (any order; None = missing) in the image order shared by the
    three ``gts`` (easy, medium, hard WiderGT).  Returns ([ap_easy, ap_medium, ap_hard], [pr_curves])."""
    norm = normalise_scores([None if p is None else sort_by_score(p) for p in preds])
    curves = [evaluate_setting(norm, g, iou_thresh, mimic_eval_bug) for g in gts]
    return [voc_ap(c[:, 1], c[:, 0]) for c in curves], curves


def wider_eval(pred_dir, gt_dir_base, silent=True, parallel=False, mimic_eval_bug=True, IoU_thresh=0.5):
    """The reference's entry point (wider_eval.py:180-222): detection text files + the toolbox's ground_truth
    directory -> (ap[3], pr_curve[3])."""
    face = load_gt_mat(os.path.join(gt_dir_base, 'wider_face_val.mat'))
    preds = read_predictions(pred_dir, face)
    gts = [load_gt_mat(os.path.join(gt_dir_base, 'wider_%s.mat' % s)) for s in SETTINGS]
    return evaluate(preds, gts, IoU_thresh, mimic_eval_bug)
