"""cfg_dump (lib/utils/get_config.py:76-77, used by train_test.py:131-132 for <output_dir>/cfgs.txt): the dumped TOML
reads back to the configuration it was written from."""
import io
import os

import numpy as np
import tomli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _plain(v):
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple, np.ndarray)):
        return [_plain(x) for x in v]
    if isinstance(v, np.generic):
        return v.item()
    return v


def test_dump_reads_back_to_the_same_dictionary():
    from smallhardface_amd import config
    config.cfg_from_file(os.path.join(ROOT, "configs", "smallhardface.toml"))
    config.cfg_from_list(["TEST.SCALES", "[100, 300]", "EXP_DIR", 'a "quoted" \\ name'])
    try:
        kept = {k: config.cfg[k] for k in config.cfg if k != "TRAIN"}   # train_test.py:132
        f = io.StringIO()
        config.cfg_dump(kept, f)
        text = f.getvalue()
        back = tomli.loads(text)
        assert back == _plain(kept)
        assert "TRAIN" not in back and back["TEST"]["SCALES"] == [100, 300]
        # sorted keys, a table's scalars before its sub-tables (what `toml.dump(_sort_dict(cfg))` writes)
        top = [l.split(" = ")[0] for l in text.split("\n[")[0].splitlines() if " = " in l]
        assert top == sorted(top)
        heads = [l for l in text.splitlines() if l.startswith("[")]
        assert "[TEST]" in heads and all(h.startswith("[TEST") or h.count(".") == 0 for h in heads if "TEST" in h)
    finally:
        config.cfg_reset()
