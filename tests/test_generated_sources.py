"""Generated sources are committed; this checks they match their generators."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fir_mac_inc_matches_its_generator():
    spec = importlib.util.spec_from_file_location("gen_fir_mac", os.path.join(ROOT, "tools", "gen_fir_mac.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    want = gen.whole_file()
    have = open(os.path.join(ROOT, "iridium-sniffer_amd", "csrc", "fir_mac.inc")).read()
    assert have == want, "iridium-sniffer_amd/csrc/fir_mac.inc is stale: run python tools/gen_fir_mac.py"
    # every group adds its taps in ascending order and never fuses: 2N multiplies, 2N adds, no fma
    for n in range(1, 7):
        body = gen.gen(n)
        assert body.count("v_pk_mul_f32") == 2 * n and body.count("v_pk_add_f32") == 2 * n and "fma" not in body
    # the two-chain groups of the register-resident decimator: per chain 2N multiplies and 2N adds, taps ascending
    for n in (1, 2, 4):
        body = gen.gen2(n)
        assert body.count("v_pk_mul_f32") == 4 * n and body.count("v_pk_add_f32") == 4 * n and "fma" not in body
