"""python -m plssvm_amd.predict == plssvm-predict (src/main_predict.cpp)."""
import sys

from .cli import predict_main

if __name__ == "__main__":
    sys.exit(predict_main())
