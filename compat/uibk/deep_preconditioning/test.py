"""`uibk.deep_preconditioning.test` (test.py:31-221): the benchmark harness and its entry point."""
from deeppreconditioning_amd.benchmark_suite import BenchmarkSuite, main  # noqa: F401


if __name__ == "__main__":
    main()
