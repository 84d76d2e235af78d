"""Entry-point shim so `import opts; opts.parse_opt()` works from the repository root exactly as in
the reference checkout (opts.py:3)."""
from cooperativeimagecaptioning_amd.opts import parse_opt, build_parser  # noqa: F401
