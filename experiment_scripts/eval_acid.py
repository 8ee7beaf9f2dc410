"""Evaluation on ACID (mirrors reference experiment_scripts/eval_acid.py): the loop of eval_realestate10k.py over the ACID download —
the reference's ACIDVis (dataset/acid_dataio.py:504-) is the same reader as RealEstate10kVis pointed at other directories
(eval_acid.py: img_root "data_download/acid/test", pose_root "poses/acid/test.mat")."""
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402
from eval_realestate10k import evaluate  # noqa: E402

if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(functools.partial(evaluate, default_data=("data_download/acid/test", "poses/acid/test.mat")), opt)
