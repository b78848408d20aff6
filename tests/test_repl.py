"""The command surface (raytracing_amd/rtgi_hip_repl, grammar of driver/interaction.cpp:57-406 in the reference tree
"Assignment 3/rtgi-2021-a03-ref/") without a GPU: --dry-run parses and checks every command, loads the scene files,
and refuses what the reference refuses. The scripts below are this repository's own; where the reference tree is
present (this container only) its scripts for the scenes that exist are parsed too."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPL = os.path.join(ROOT, "raytracing_amd", "rtgi_hip_repl")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference/Assignment 3/rtgi-2021-a03-ref"

FULL_SCRIPT = """
# every command of the grammar
at -2.4008 1.7537 -0.050779
look 0.85235 -0.52249 -0.022323
up 0 1 0
camup 0.52233 0.85264 -0.013184
default-brdf layered-gtr2
load render-data/brdf-test-noal.obj
load render-data/tri.obj second
raytracer hip indexed esc devices 1
bvh sah 16
bvh triangles multiple 4
hip reorder off
hip sharding samples
hip tile 32
hip seed 7
hip jitter off
commit
algo local
sppx 16
bounces 2
pointlight pos 2 1.5 0 col 4 2 1
pointlight replace pos 2 1.5 0 col 4 2 1
resolution 1280 720
outfile frame.png
material list
mesh list
material select microfacet
material ior 2.4
material roughness 0.025
material texture drop
material albedo 0.5 0.5 0.5
material emissive 0 0 0
material brdf phong
material show
material select second/tri
omp off
omp on
stats clear
stats print
echo hello there
at
history
run
rt_bench
"""


def _run(script_text, tmp_path, cwd=GOLDEN, extra=()):
    path = tmp_path / "script"
    path.write_text(script_text)
    return subprocess.run([REPL, "--dry-run", "--strict", "-s", str(path), *extra], cwd=cwd, capture_output=True, text=True, timeout=60)


def test_every_command_parses(tmp_path):
    r = _run(FULL_SCRIPT, tmp_path)
    assert r.returncode == 0, r.stderr
    assert "hello there" in r.stdout and "microfacet.002" in r.stdout and "second/tri" in r.stdout
    assert "brdf       phong" in r.stdout and "ior        2.4" in r.stdout
    assert "vec3(-2.4008, 1.7537, -0.050779)" in r.stdout  # `at` without arguments prints the position


@pytest.mark.parametrize("script,message", [
    ("at 1 2\n", "requires 3 numerical components"),
    ("up 0 1 0\nup 0 0 1\n", "Cannot set scene up vector twice"),
    ("load no-such-file.obj\n", "Failed to load file"),
    ("algo radiosity\n", "There is no gi algorithm called 'radiosity'"),
    ("raytracer embree\n", "There is no ray tracer called 'embree'"),
    ("raytracer bbvh esc\n", "This combination is technically problematic"),
    ("commit\n", "There is no scene data to work with"),
    ("load render-data/tri.obj\ncommit\n", "There is no ray traversal scheme"),
    ("load render-data/tri.obj\nraytracer hip\ncommit\nrun\n", "prior to running"),
    ("load render-data/tri.obj\nraytracer hip\ncommit\nraytracer hip indexed\nalgo primary\nrun\n", "up-to-date acceleration structure"),
    ("raytracer hip\nload render-data/tri.obj\ncommit\nload render-data/2tri.obj\nalgo primary\nrun\n", "out-dated"),
    ("load render-data/tri.obj\nraytracer hip\ncommit\nalgo local\nrun\n", "There is no light"),
    ("material roughness 0.5\n", "No material selected"),
    ("load render-data/tri.obj\nmaterial select nothing\n", "No material called 'nothing'"),
    ("default-brdf velvet\n", "No such brdf defined: velvet"),
    ("load render-data/tri.obj\nmaterial select tri\nmaterial texture no-such.png\n", "cannot open no-such.png"),
    ("load render-data/tri.obj\nmaterial select tri\nmaterial texture render-data/tri.obj\n", "is not a PNG file"),
    ("pointlight at 1 2 3\n", "poinlight [replace] pos x y z col x y z"),
    ("sppx many\n", "positive integral value"),
    ("resolution 640\n", "requires 2 integral values"),
    ("stats reset\n", "No such stats subcommand"),
    ("frobnicate\n", "Unknown command"),
    ("bvh sah 16\n", "Unknown command"),  # no tracer yet to take it
])
def test_refusals(script, message, tmp_path):
    r = _run(script, tmp_path)
    assert r.returncode == 1
    assert message in r.stderr, r.stderr


def test_errors_do_not_stop_the_script(tmp_path):
    r = _run("frobnicate\necho still here\n", tmp_path)
    assert "still here" in r.stdout and "command 1 (frobnicate): Unknown command" in r.stderr


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only present in the build container")
@pytest.mark.parametrize("name", ["a1-tri", "a2-tris", "a3-brdf-test"])
def test_reference_scripts_parse_unchanged(name):
    r = subprocess.run([REPL, "--dry-run", "--strict", "-s", os.path.join("scripts", name)], cwd=REFERENCE, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
