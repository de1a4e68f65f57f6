"""YAML front end of the native drivers (host logic only)."""
import pytest

from miniweatherml_amd import driver

GOOD = """
sim_time: 12.5
nens   : 2
nx_glob: 24
ny_glob: 16
nz     : 12
xlen: 12000
ylen: 8000
zlen: 20000
init_data: supercell
out_prefix: run1
dt_gcm: 900
dt_phys: 0.
out_freq: -1
"""


def test_reads_the_reference_keys_with_their_defaults(tmp_path):
    p = tmp_path / "in.yaml"
    p.write_text(GOOD)
    c = driver.load_config(str(p))
    assert (c["nx_glob"], c["ny_glob"], c["nz"], c["nens"]) == (24, 16, 12, 2)
    assert c["sim_time"] == 12.5 and c["dt_phys"] == 0.0 and c["out_freq"] == -1.0 and c["init_data"] == "supercell"
    assert c["enable_gravity"] is True and c["file_per_process"] is False          # .as<bool>(true) / .as<bool>(false)
    p.write_text(GOOD + "enable_gravity: false\nkeras_weights_txt: w.txt\n")
    c = driver.load_config(str(p))
    assert c["enable_gravity"] is False and c["keras_weights_txt"] == "w.txt"


def test_rejects_bad_input(tmp_path):
    p = tmp_path / "in.yaml"
    p.write_text(GOOD.replace("nz     : 12\n", ""))
    with pytest.raises(KeyError, match="nz"):
        driver.load_config(str(p))
    p.write_text("- just\n- a list\n")
    with pytest.raises(ValueError, match="Invalid YAML"):
        driver.load_config(str(p))
    with pytest.raises(ValueError, match="unknown experiment"):
        driver.run("no_such_experiment", str(p))
    assert driver.main(["supercell_example", str(tmp_path / "missing.yaml")]) == 2
