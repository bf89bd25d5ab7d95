"""Where does the file-to-detections loop lose time against the same images from memory?  (GPU box)"""
import os, sys, time, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from PIL import Image
from smallhardface_amd import caffe, prototxt as P, weights, test as T
from smallhardface_amd.config import cfg, cfg_from_file
cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
caffe.set_mode_gpu(); caffe.set_device(0)
msg = P._add_dimension_reduction(P.build_test_template(True))
params = weights.synth_params(msg, seed=1234)
net = caffe.Net(None, prototxt_text=P.dumps(msg))
for name, blobs in params.items():
    for i, arr in enumerate(blobs):
        net.params[name][i].data[...] = arr
net.commit_params(); net.set_conv_mode("f16x3")
shapes = [(768, 1024), (683, 1024), (1024, 732), (1365, 1024), (576, 1024), (1024, 819), (1536, 1024), (1024, 1024)]
rng = np.random.default_rng(4242)
tdir = tempfile.mkdtemp()
paths = []
for k in range(32):
    h, w = shapes[k % 8]
    low = rng.integers(0, 256, (h // 16 + 2, w // 16 + 2, 3)).astype(np.uint8)
    im = np.asarray(Image.fromarray(low).resize((w, h), Image.BICUBIC)).astype(np.int16)
    im = np.clip(im + rng.integers(-12, 13, im.shape), 0, 255).astype(np.uint8)
    paths.append(os.path.join(tdir, "i%02d.jpg" % k)); Image.fromarray(im).save(paths[-1], quality=90)
fd = T.FusedDetector(net, n_lanes=10, mode="group")
dp = T.DevicePyramid(net, n_slots=2)
T.fused_image_loop(net, paths, fd=fd, dp=dp)
mem = [T._imread(p) for p in paths]
def from_mem():
    t0 = time.perf_counter(); sub = col = 0.0
    for im in mem:
        a = time.perf_counter()
        fd.submit(dp.units(im, net=fd.next_head()), 0.05, on_device=True)
        b = time.perf_counter(); sub += b - a
        if fd.pending() > 1:
            fd.collect(); col += time.perf_counter() - b
    while fd.pending(): fd.collect()
    dt = time.perf_counter() - t0
    return 32 / dt, 1e3 * sub / 32, 1e3 * col / 32
for rep in range(3):
    if rep == 2:
        sys.setswitchinterval(0.0005)
        print("-- switch interval 0.5 ms", flush=True)
    print("memory      : %.1f img/s submit %.2f collect %.2f" % from_mem(), flush=True)
    for pf in (0, 1, 2, 2, 2):
        st = {}
        t0 = time.perf_counter()
        T.fused_image_loop(net, paths, fd=fd, dp=dp, prefetch=pf, stats=st)
        dt = time.perf_counter() - t0
        print("prefetch %d  : %.1f img/s" % (pf, 32 / dt), {k: round(v, 2) for k, v in st.items()}, flush=True)
