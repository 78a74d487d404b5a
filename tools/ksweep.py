import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_pp import run
run([(7680, 8192, 2048), (7680, 2048, 2048), (7680, 2048, 5888), (7680, 11776, 2048), (2560, 8192, 2048), (8192, 8192, 8192)], [5], rounds=7)
