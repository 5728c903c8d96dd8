"""Command-line entry of the training loop -- the role of the reference's main.py / MAPPO_parallel_main.py.

    python -m distributed_multi_agent_reinforcement_learning_amd.main --config cfg3 --iterations 10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m distributed_multi_agent_reinforcement_learning_amd.main

`--config` is one of the BASELINE configurations (cfg1..cfg3) or the path of a reference-schema config.yaml;
dotted overrides follow as KEY=VALUE (e.g. runtime.num_envs=1024 algo.depth=1).
"""
import argparse
import ast

from .config import baseline_config, load_config
from .trainer import train_agent_multiprocessing


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--iterations", type=int, default=None, help="stop after this many iterations (default: max_train_steps)")
    ap.add_argument("--eval-envs", type=int, default=64)
    ap.add_argument("--eval-every", type=int, default=1)
    ap.add_argument("overrides", nargs="*", help="dotted overrides KEY=VALUE")
    args = ap.parse_args(argv)
    ov = {}
    for item in args.overrides:
        k, _, v = item.partition("=")
        try:
            ov[k] = ast.literal_eval(v)
        except (ValueError, SyntaxError):
            ov[k] = v
    cfg = baseline_config(args.config, **ov) if args.config in ("cfg1", "cfg2", "cfg3", "cfg4") else load_config(args.config, **ov)
    train_agent_multiprocessing(cfg, max_iterations=args.iterations, num_eval_envs=args.eval_envs, eval_every=args.eval_every)


if __name__ == "__main__":
    main()
