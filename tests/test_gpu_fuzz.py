"""A short run of the randomised differential test (tests/fuzz_vs_oracle.py): random scenes, cameras, windows, bounce limits,
search modes and dispatch sequences, HIP against the CPU oracle bit for bit."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2])
def test_random_cases_against_the_oracle(seed):
    import fuzz_vs_oracle
    assert fuzz_vs_oracle.main(["--cases", "20", "--seed", str(seed)]) == 0
