"""Write an Erdős–Rényi METIS file: python scratch/experiments/write_er_metis.py n m seed path"""
import sys
sys.path.insert(0, ".")
from tools import graphgen as gg  # noqa: E402
n, m, seed, path = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
open(path, "w").write(gg.metis_text(gg.erdos_renyi(n, m, seed)))
