"""Detection output writers (the consumers that define "0-pixel" parity) and a minimal imdb.

  write_detections_wider   lib/datasets/wider.py:143-167   '%d %d %d %d %g' with int() truncation
  write_detections_fddb    lib/datasets/fddb.py:57-73      '{:.3f} {:.3f} {:.3f} {:.3f} {:.10f}' (w,h = +1)
  write_detections_afw     lib/datasets/afw.py:45-60       (same layout for pascalface.py:45-60)
  ImageList                the part of lib/datasets/imdb.py the inference driver touches
                           (image_path_at, num_classes, name, __len__, evaluate_detections)

The WIDER evaluator itself (lib/wider_eval_tools/wider_eval.py) needs the ground-truth .mat files,
which are not in the reference tree: out of scope here.
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


class ImageList(object):
    """The imdb surface lib/test.py uses, over a plain list of image paths."""

    def __init__(self, name, image_paths, writer=write_detections_wider, root=''):
        self.name = name
        self._image_paths = list(image_paths)
        self._root = root
        self._classes = ['bg', 'face']
        self._writer = writer

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
        return 'detections written to {}'.format(out)
