"""Launcher for the reference's own scripts on the drop-in modules:

    python -m point2cyl_amd.dropin.run /path/to/point2cyl/train_Point2Cyl_without_sketch.py --pred_seg --pred_normal --pred_bb ...
    python -m point2cyl_amd.dropin.run /path/to/point2cyl/eval.py --logdir <run> --is_visu

`python script.py` puts the SCRIPT's directory at sys.path[0], in front of PYTHONPATH - and losses.py, data_utils.py, global_variables.py and
the `models` package live in that directory, so a PYTHONPATH entry alone can only shadow `pointnet_extrusion` (found through the
`models/` directory the scripts APPEND, train_Point2Cyl_without_sketch.py:14-16).  This launcher builds the path the other way round -
drop-in directories first, then the script's directory - and runs the script as `__main__` unchanged (runpy.run_path does not touch sys.path
for a plain file).  Names the drop-ins do not override fall through to the reference modules further down the path (_shadow.py)."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def dropin_path(script_dir=None):
    """The sys.path prefix that makes the reference's import names resolve to the drop-ins."""
    p = [HERE, os.path.join(HERE, "models"), ROOT]
    if script_dir:
        p.append(script_dir)
    return p


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        sys.stderr.write(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    sys.path[:] = dropin_path(os.path.dirname(script)) + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in
                                                          (HERE, os.path.join(HERE, "models"), ROOT, os.path.dirname(script))]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
