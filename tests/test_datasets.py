"""Detection writers against the reference's line formats (tests/golden/write_detections.json)."""
import json
import os

import numpy as np

from smallhardface_amd import datasets as D

GOLD = os.path.join(os.path.dirname(__file__), "golden", "write_detections.json")


def test_wider_line_format():
    g = json.load(open(GOLD))
    for row, line in zip(g["rows"], g["lines"]):
        assert D.wider_line(np.array(row)) == line
    # truncation, not rounding; widths from truncated corners
    assert D.wider_line(np.array([10.9, 20.2, 50.99, 80.5, 0.5])) == '10 20 40 60 0.5 \n'


def test_writers_roundtrip(tmp_path):
    paths = ["0--Parade/a.jpg", "1--Handshaking/b.jpg"]
    boxes = [[], [np.array([[1.5, 2.5, 11.0, 22.0, 0.9], [3, 4, 5, 6, 0.25]]), np.zeros((0, 5))]]
    D.write_detections_wider(paths, boxes, str(tmp_path))
    t = open(tmp_path / "0--Parade" / "a.txt").read().splitlines()
    assert t[0] == paths[0] and t[1] == "2" and t[2] == "1 2 10 20 0.9 " and t[3] == "3 4 2 2 0.25 "
    assert open(tmp_path / "1--Handshaking" / "b.txt").read().splitlines()[1] == "0"
    D.write_detections_fddb(paths, boxes, str(tmp_path / "fddb"))
    f = open(tmp_path / "fddb" / "detection_rect.txt").read().splitlines()
    assert f[0] == "0--Parade/a" and f[1] == "2" and f[2] == "1.500 2.500 10.500 20.500 0.9000000000"
    D.write_detections_afw(paths, boxes, str(tmp_path / "afw"))
    a = open(tmp_path / "afw" / "afw_res.txt").read().splitlines()
    assert a[0] == "a 0.900 1.5 6.6 11.0 22.0"
    imdb = D.ImageList("toy", paths)
    assert len(imdb) == 2 and imdb.num_classes == 2 and imdb.image_path_at(1) == paths[1]
    assert "written" in imdb.evaluate_detections(boxes, str(tmp_path / "ev"))
