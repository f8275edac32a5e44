"""test infrastructure (by hand): optimize() through its warm start (csrc/warm.hip + the stream-K Gram kernel), the same call many
times -- any number of distinct results other than 1 is a data race or an uninitialised read."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from race_hunt import hunt
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hunt("fw", 20000, 512, 400, "float32", reps)         # k ~ 400 <= d: the warm start is the answer
hunt("fw", 8000, 256, 700, "float32", reps)          # k > d: dependent columns, pivots after the warm start
hunt("giga", 6000, 200, 500, "float32", reps)
hunt("fw", 20000, 1024, 1300, "float32", reps // 2)  # k ~ 1300: three-block-wide tiles of the Gram kernel split over workgroups
