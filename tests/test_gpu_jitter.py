"""The octet pair kernel's K / D / E hand-shakes under perturbed timing, and its counted wait without the counting
(ADVICE r3).  Three one-team-size builds of the library, each compared bit for bit with the 16-lane step kernel of the same build
(tests/jitter_child.py):
  jitter_n3   -DCS_JITTER        pseudo-random pauses of up to two thirds of a step in every role at every counter read / post: the
                                 wavefronts meet in interleavings the natural timing never produces;
  odsafe_n3   -DCS_OD_SAFE_WAIT  D's wait for its asynchronous requests (a row of MT19937 words, a reset's first attempt batch, both
                                 loaded straight into LDS a step ahead) is `s_waitcnt vmcnt(STEP_STORES)` in the shipped kernel -- it
                                 rests on vector-memory operations retiring in order and on the number of stores issued after the
                                 requests; here it is a full drain;
  odsync_n3   -DCS_OD_ASYNC=0    no requests ahead of time at all: every row and every attempt batch is loaded where it is used.
If the shipped kernel ever read LDS before a request had landed, it would differ from the step kernel where these two do not.
The three-wavefront variant's row refreshes run in its emitting wavefront (request / old tape meanwhile / adoption, CS_OD_E_REFRESH):
the jitter build pauses at those hand-shakes too, and the child's last scenario makes every env refresh its row every few steps.
  jitter_n5   -DCS_JITTER        (round 6; replaces round 5's build of the 5-lane packing, which is gone) the jitter build for teams of
                                 5: the pair kernels of BASELINE configs 3 and 5, whose detection pass tests in packed fp32 first.
  prewide_n5  -DCS_PREFILTER_EPS_SCALE=100.0f -DCS_OD_PREFILTER=2 -DCS_OCT_PREFILTER=1  (round 6) that pre-filter in ALL octet kernels (the
                                 shipped build has it in the pair variant only) with its fallback band a hundred times wider, so that
                                 the fp64 redo runs in a tenth of the wavefront-steps instead of one in two thousand."""
import concurrent.futures
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = {"jitter_n3": ["-DCS_JITTER"], "odsafe_n3": ["-DCS_OD_SAFE_WAIT"], "odsync_n3": ["-DCS_OD_ASYNC=0"],
            # (round 6) the jitter build for teams of 5 (BASELINE configs 3 / 5 run the 5-agent pair kernels)
            "jitter_n5": ["-DCS_JITTER"],
            # (round 6) the sensor pre-filter's band a hundred times wider: a tenth of the wavefront-steps redo the pass in fp64, the rest
            # take the packed-fp32 verdict -- both paths of oct_detect_impl<PRE> run thousands of times against the step kernel
            "prewide_n5": ["-DCS_PREFILTER_EPS_SCALE=100.0f", "-DCS_OD_PREFILTER=2", "-DCS_OCT_PREFILTER=1"]}


def team_size(name):
    return int(name.rsplit("_n", 1)[1])


def variant_path(name):
    from cooperative_search_amd import build as b
    return b.variant_path(name)


def build_variant(name):
    """build/var/<name>.so through cooperative_search_amd.build.build_variant (also run by __graft_entry__.build(), so that the
    variants travel to the GPU box): compiled with -DCS_SOURCE_HASH=<hash of sources + flags>, current when its recorded hash is
    that -- never by mtime.  None when it is missing and there is no hipcc."""
    from cooperative_search_amd import build as b
    return b.build_variant(name, VARIANTS[name], only_n=team_size(name))


def expected_hash(name):
    from cooperative_search_amd import build as b
    return b.variant_hash(VARIANTS[name], team_size(name))


def build_all_variants():
    """All of them, side by side (each is one single-threaded hipcc run of ~2 minutes)."""
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(VARIANTS)) as ex:
        return list(ex.map(build_variant, VARIANTS))


def build_jitter_lib():   # (the name round 4's first version exported)
    return build_variant("jitter_n3")


@pytest.mark.parametrize("name", list(VARIANTS))
def test_pair_kernel_variant_builds_equal_the_step_kernel(name):
    lib = build_variant(name)
    if lib is None:
        pytest.skip(f"no {name} build and no hipcc")
    # the variant must have been compiled from the PRESENT sources with ITS flags: the hash compiled into it says so (a stale
    # build would still equal its own step kernel and pass without testing the current protocol); the child's loader checks again
    from cooperative_search_amd import build as b
    assert b.embedded_hash(lib) == expected_hash(name), f"{lib} is a stale build: run __graft_entry__.build() where hipcc is"
    env = dict(os.environ, COOPSEARCH_LIB=lib, COOPSEARCH_LIB_HASH=expected_hash(name))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "jitter_child.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert p.stdout.count("bit-identical") == (3 if name == "prewide_n5" else 2), p.stdout
