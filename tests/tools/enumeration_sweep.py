"""Global optimality at N = 10 on more problems than the suite checks: the body of
tests/test_gpu_round6.py::test_branch_and_bound_optimum_is_the_minimum_over_every_assignment_at_n10 (all P^N assignments of a problem as
pinned QPs on the GPU; the minimum is the branch and bound's cost, no earlier factor has a feasible assignment, SciPy confirms the
winner on the unreduced model) for several seeds.   PYTHONPATH=. python tests/tools/enumeration_sweep.py [seeds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # (torch before the HIP library: one HIP runtime in the process)
    torch.cuda.init()
    from faster_amd import capi
    import test_gpu_round6 as T

    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = capi.Context(0)
    t0 = time.time()
    problems = qps = 0
    for k in range(seeds):
        T.test_branch_and_bound_optimum_is_the_minimum_over_every_assignment_at_n10(ctx, seed_whole=7000 + 2 * k, seed_pairs=7001 + 2 * k)
        n, q = T._last_enumeration
        problems += n
        qps += q
        print("seeds %d / %d: %d problems, %d pinned QPs so far, every minimum = the branch and bound's cost | %d s" % (7000 + 2 * k, 7001 + 2 * k, problems, qps, time.time() - t0), flush=True)
    print("ENUMERATION SWEEP DONE: %d problems at N = 10 (P <= 3), %d pinned QPs, 0 cheaper assignments, 0 feasible earlier factors" % (problems, qps))
    ctx.close()


if __name__ == "__main__":
    main()
