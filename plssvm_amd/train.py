"""python -m plssvm_amd.train == plssvm-train (src/main_train.cpp)."""
import sys

from .cli import train_main

if __name__ == "__main__":
    sys.exit(train_main())
