"""Entry point kept for drop-in parity with the reference's experiment_scripts/render_unposed_traj.py: same flags, same render loop as the
RealEstate10K script (the reference scripts differ only in the dataset / pose source, which is not built yet)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402
from render_realestate10k_traj import render as run  # noqa: E402

if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(run, opt)
