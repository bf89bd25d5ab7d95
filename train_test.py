#!/usr/bin/env python3
"""Entrance for testing -- the inference half of the reference's train_test.py (:32-137):

    python train_test.py --train false --conf configs/smallhardface.toml \\
        --amend TEST.MODEL final.caffemodel DATA_DIR /data/WIDER TEST.GPU_ID "[0,1,2,3]"

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 train_test.py --train false \\
        --conf configs/smallhardface.toml --amend TEST.MODEL final.caffemodel TEST.SHARD pyramid
        (not in the reference: every image's PYRAMID sharded over the ranks, smallhardface_amd/pyramid.py ShardedDetector)

Same flags and control flow (cfg_from_file -> NO_CACHE -> --amend -> seed -> manipulate_test ->
test_net).  ``--train true`` is refused: training is outside this build's scope.  The image
database is a plain list: ``<DATA_DIR>/<TEST.DB>.txt`` (one image path per line, relative to
DATA_DIR) or, failing that, every *.jpg under ``<DATA_DIR>/images``.
"""
import argparse
import datetime
import glob
import logging
import os
import os.path as osp
import sys

import numpy as np

from smallhardface_amd.config import cfg, cfg_dump, cfg_from_file, cfg_from_list, get_output_dir
from smallhardface_amd.datasets import ImageList
from smallhardface_amd.prototxt import manipulate_test
from smallhardface_amd.test import dist_env, test_net

logging.basicConfig(format='%(asctime)s,%(msecs)d %(levelname)-8s [%(filename)s:%(lineno)d] %(message)s',
                    datefmt='%m-%d-%Y:%H:%M:%S',
                    level=logging.DEBUG if os.environ.get('DEBUG') == '1' else logging.INFO)
logger = logging.getLogger(__name__)


def parser():
    p = argparse.ArgumentParser('Train and test', description='Give settings')
    p.add_argument('--train', dest='train', help='do training', default='true')
    p.add_argument('--test', dest='test', help='do testing', default='true')
    p.add_argument('--conf', dest='conf_file', help='provide configure file', default='')
    p.add_argument('--amend', dest='set_cfgs', help='provide amend cfgs', default=None, nargs=argparse.REMAINDER)
    return p.parse_args()


def get_imdb(name):
    lst = osp.join(cfg.DATA_DIR, name + '.txt')
    if osp.isfile(lst):
        paths = [l.strip() for l in open(lst) if l.strip()]
    else:
        paths = sorted(glob.glob(osp.join(cfg.DATA_DIR, 'images', '**', '*.jpg'), recursive=True))
        paths = [osp.relpath(p, cfg.DATA_DIR) for p in paths]
    if not paths:
        raise IOError('no image list {} and no jpgs under {}/images'.format(lst, cfg.DATA_DIR))
    return ImageList(name, paths, root=cfg.DATA_DIR, ground_truth=osp.join(cfg.DATA_DIR, 'ground_truth'))


if __name__ == '__main__':
    args = parser()
    if args.conf_file:
        cfg_from_file(args.conf_file)
    cfg.TEST.NO_CACHE = True  # train_test.py:58
    if args.set_cfgs:
        cfg_from_list(args.set_cfgs)
    np.random.seed(cfg.RNG_SEED)
    if args.train.lower() == 'true':
        sys.exit('training is outside the scope of this build (inference hot path only): pass --train false')
    if args.test.lower() == 'true':
        cfg.NAME_TIME = datetime.datetime.now().strftime('%Y%m%d_%H%M%S')
        imdb = get_imdb(cfg.TEST.DB)
        rank, world, _ = dist_env()
        if world > 1 and rank != 0:
            # TEST.SHARD pyramid under torch.distributed.run: rank 0 owns the results directory, the other ranks keep their
            # stderr.log / test.prototxt in a sibling of their own (the ranks' clocks need not agree on NAME_TIME)
            cfg.NAME_TIME += '_rank%d' % rank
        output_dir = get_output_dir(cfg.TEST.DB, cfg.NAME_TIME)
        # train_test.py:122-124: from here on the run's stderr goes to <output_dir>/stderr.log (warnings, tracebacks)
        f = open(osp.join(output_dir, 'stderr.log'), 'w', 1)
        sys.stderr.flush()
        os.dup2(f.fileno(), sys.stderr.fileno())
        target_test = osp.join(output_dir, 'test.prototxt')
        manipulate_test(cfg.TEST.PROTOTXT, target_test)
        with open(osp.join(output_dir, 'cfgs.txt'), 'w') as cf:   # train_test.py:131-132
            cfg_dump({i: cfg[i] for i in cfg if i != 'TRAIN'}, cf)
        test_net(imdb, output_dir, target_test, no_cache=cfg.TEST.NO_CACHE)
        f.close()
