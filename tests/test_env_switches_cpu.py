"""Every environment switch the library or the plugin reads is listed in INTEGRATION.md (section 4a / 2b): a hook that exists only
in the source is a hook nobody can use — and one that silently changes a route is a surprise for the next maintainer."""
import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def _switches():
    names = set()
    for sub in ("infinitensor_amd/csrc", "infinitensor_amd/plugin/src", "infinitensor_amd/plugin/include"):
        for f in (REPO / sub).rglob("*"):
            if f.suffix in (".hip", ".h", ".cc", ".inc", ".hpp"):
                txt = f.read_text(errors="replace")
                names.update(re.findall(r'getenv\("([A-Z0-9_]+)"\)', txt))
                names.update(re.findall(r'env_int\("([A-Z0-9_]+)"', txt))
                names.update(re.findall(r'envOn\("([A-Z0-9_]+)"\)', txt))
    return sorted(n for n in names if n.startswith(("IROCM_", "INFINI_ROCM_")))


def test_every_environment_switch_is_documented():
    doc = (REPO / "INTEGRATION.md").read_text()
    # families written with a wildcard or braces in the document: INFINI_ROCM_FUSE_*, INFINI_ROCM_GROUP_*, INFINI_ROCM_FUSE_{A,B,...}
    families = [m.rstrip("*") for m in re.findall(r"(INFINI_ROCM_[A-Z_]*\*)", doc)]
    for prefix, body in re.findall(r"(INFINI_ROCM_[A-Z_]*)\{([A-Z_,]+)\}", doc):
        families += [prefix + part for part in body.split(",")]
    missing = [n for n in _switches() if n not in doc and not any(n.startswith(fam) for fam in families)]
    assert not missing, f"environment switches read by the sources but absent from INTEGRATION.md: {missing}"
