"""python -m plssvm_amd.scale == plssvm-scale (src/main_scale.cpp)."""
import sys

from .cli import scale_main

if __name__ == "__main__":
    sys.exit(scale_main())
