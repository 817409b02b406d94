"""The every-bin mask checker lives in oracle/maskcheck.py (bench.py's parity_check uses it too); the GPU tests import it from
here."""
from oracle.maskcheck import EPS_R, check_masked  # noqa: F401
