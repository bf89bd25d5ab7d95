"""Detection output writers (the consumers that define "0-pixel" parity) and a minimal imdb.

  write_detections_wider   lib/datasets/wider.py:143-167   '%d %d %d %d %g' with int() truncation
  write_detections_fddb    lib/datasets/fddb.py:57-73      '{:.3f} {:.3f} {:.3f} {:.3f} {:.10f}' (w,h = +1)
  write_detections_afw     lib/datasets/afw.py:45-60       one line per box, ymin lowered by 20 % of the height
  write_detections_pascal  lib/datasets/pascalface.py:45-60  the same layout in pascal_res.txt
  ImageList                the part of lib/datasets/imdb.py the inference driver touches
                           (image_path_at, num_classes, name, __len__, evaluate_detections)
  writer_for(db)           imdb name (cfg.TEST.DB) -> writer, like lib/datasets/factory.py picks the class

All four are pinned by tests/golden/writers.json (files written by the reference's own classes).  The WIDER
evaluator is smallhardface_amd/wider_eval.py; ImageList runs it when a ground_truth directory is given.
"""
import os


def write_detections_wider(image_paths, all_boxes, output_dir='./output/'):
    """``all_boxes[1][i]`` = (n,5) detections of image i (class 1 = face), like test_net returns."""
    for i, img_path in enumerate(image_paths):
        img_name = os.path.basename(img_path)
        img_dir = img_path[:img_path.find(img_name) - 1]
        res_dir = os.path.join(output_dir, img_dir)
        os.makedirs(res_dir, exist_ok=True)
        with open(os.path.join(res_dir, img_name.replace('jpg', 'txt')), 'w') as f:
            f.write(img_path + '\n')
            f.write(str(len(all_boxes[1][i])) + '\n')
            for det in all_boxes[1][i]:
                f.write(wider_line(det))


def wider_line(det):
    return '%d %d %d %d %g \n' % (int(det[0]), int(det[1]), int(det[2]) - int(det[0]),
                                  int(det[3]) - int(det[1]), det[4])


def write_detections_fddb(image_paths, all_boxes, output_dir='./output/'):
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, 'detection_rect.txt'), 'w') as f:
        for i, img_path in enumerate(image_paths):
            f.write('{:s}\n'.format(os.path.splitext(img_path)[0]))
            dets = all_boxes[1][i]
            f.write('{:d}\n'.format(dets.shape[0]))
            for d in dets:
                f.write('{:.3f} {:.3f} {:.3f} {:.3f} {:.10f}\n'.format(d[0], d[1], d[2] - d[0] + 1, d[3] - d[1] + 1, d[4]))


def write_detections_afw(image_paths, all_boxes, output_dir='./output/', fname='afw_res.txt'):
    os.makedirs(output_dir, exist_ok=True)
    with open(os.path.join(output_dir, fname), 'w') as f:
        for i, img_path in enumerate(image_paths):
            img_name = os.path.splitext(os.path.basename(img_path))[0]
            for res in all_boxes[1][i]:
                xmin, ymin, xmax, ymax = res[:4]
                ymin += 0.2 * (ymax - ymin + 1)
                f.write('{:s} {:.3f} {:.1f} {:.1f} {:.1f} {:.1f}\n'.format(img_name, res[-1], xmin, ymin, xmax, ymax))


def write_detections_pascal(image_paths, all_boxes, output_dir='./output/'):
    write_detections_afw(image_paths, all_boxes, output_dir, fname='pascal_res.txt')


def writer_for(db_name):
    """cfg.TEST.DB ('wider_val', 'fddb_val', 'afw_val', 'pascalface_val') -> its detection writer."""
    for key, w in (('wider', write_detections_wider), ('fddb', write_detections_fddb), ('afw', write_detections_afw),
                   ('pascalface', write_detections_pascal)):
        if db_name.startswith(key):
            return w
    raise KeyError('Unknown dataset: {}'.format(db_name))


class ImageList(object):
    """The imdb surface lib/test.py uses, over a plain list of image paths."""

    def __init__(self, name, image_paths, writer=None, root='', ground_truth=None):
        self.name = name
        self._image_paths = list(image_paths)
        self._root = root
        self._classes = ['bg', 'face']
        if writer is None:
            try:
                writer = writer_for(name)
            except KeyError:
                writer = write_detections_wider
        self._writer = writer
        self._ground_truth = ground_truth   # WIDER: directory with wider_{face,easy,medium,hard}_val.mat

    def __len__(self):
        return len(self._image_paths)

    @property
    def num_classes(self):
        return len(self._classes)

    def image_path_at(self, i):
        return os.path.join(self._root, self._image_paths[i]) if self._root else self._image_paths[i]

    def evaluate_detections(self, all_boxes, output_dir='./output/', method_name='smallhard', step=0):
        out = os.path.join(output_dir, 'detections')
        self._writer(self._image_paths, all_boxes, out)
        if self._writer is write_detections_wider and self._ground_truth and \
                os.path.exists(os.path.join(self._ground_truth, 'wider_face_val.mat')):
            # lib/datasets/wider.py:169-195 (the tarball / tensorboard side effects are not reproduced)
            from .config import cfg
            from .wider_eval import wider_eval
            ap, _ = wider_eval(out, self._ground_truth, mimic_eval_bug=cfg.MISC.MIMIC_EVAL_BUG,
                               IoU_thresh=cfg.TEST.IOU_THRESH)
            return 'Easy: {:.4f}, Medium: {:.4f}, Hard: {:.4f}'.format(*ap)
        return 'detections written to {}'.format(out)
