"""ORACLE -- test infrastructure, NOT the product.

Location of the CPU restatement's shared library (same C-ABI symbols as the HIP product library,
include/vo_hip.h + include/myslam_c.h).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this package; nothing under rgbd_visualodometry_amd/ does.
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(HERE, "_build", "liboracle_vo.so")
RUN_VO_ORACLE = os.path.join(HERE, "_build", "run_vo_oracle")
