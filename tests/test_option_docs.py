"""Every option key gnnvc_set_option accepts (single engines and the multi-device handle's own) is described in include/gnnvc.h —
the header is the ABI's documentation, and an option nobody can look up is a knob nobody can trust."""
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _set_option_keys():
    src = (ROOT / "gnn-mwvc_amd" / "csrc" / "gnnvc_engine.cpp").read_text()
    body = src[src.index("int gnnvc_set_option("):src.index("int gnnvc_get_info(")]
    keys = set(re.findall(r'k == "([a-z0-9_]+)"', body))
    multi = (ROOT / "gnn-mwvc_amd" / "csrc" / "gnnvc_multi.cpp").read_text()
    mbody = multi[multi.index("int multi_set_option("):]
    mbody = mbody[:mbody.index("\nint multi_upload(")]
    keys |= set(re.findall(r'k == "([a-z0-9_]+)"', mbody))
    return sorted(keys)


def test_every_settable_option_is_in_the_header():
    header = (ROOT / "include" / "gnnvc.h").read_text()
    keys = _set_option_keys()
    assert len(keys) > 40
    missing = [k for k in keys if f'"{k}"' not in header]
    assert not missing, missing
