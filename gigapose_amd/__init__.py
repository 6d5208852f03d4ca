"""gigapose_amd: GigaPose's coarse-pose hot path on MI355X (see README.md / DESIGN.md)."""
import os

# Kernel arguments in device memory (the HIP runtime reads this when it is loaded, i.e. it only takes effect when this package
# is imported before torch): ~210 dependent launches per step each start 1-2 us earlier -- bench.py, DESIGN.md section 6.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
