"""Drop-in for the reference's IGR/network.py: the implicit decoder and the sketch encoder on this package's kernels
(point2cyl_amd/implicit.py, point2cyl_amd/sketch.py), plus the host-only learning-rate schedule helper the with-sketch trainer takes
from the same module (IGR/network.py:176-198 with IGR/general.py:65-77)."""
from point2cyl_amd.implicit import ImplicitNet, add_latent, gradient  # noqa: F401
from point2cyl_amd.sketch import PointNetEncoder  # noqa: F401


class StepLearningRateSchedule:
    """initial * factor ** (epoch // interval), never below 5e-6 (IGR/general.py:70-77)."""

    def __init__(self, initial, interval, factor):
        self.initial, self.interval, self.factor = initial, interval, factor

    def get_learning_rate(self, epoch):
        return max(self.initial * (self.factor ** (epoch // self.interval)), 5.0e-6)


def get_learning_rate_schedules(schedule_specs):
    """[{"Type": "Step", "Initial": .., "Interval": .., "Factor": ..}, ...] -> schedule objects (IGR/network.py:176-198)."""
    out = []
    for spec in schedule_specs:
        if spec["Type"] != "Step":
            raise Exception('no known learning rate schedule of type "{}"'.format(spec["Type"]))
        out.append(StepLearningRateSchedule(spec["Initial"], spec["Interval"], spec["Factor"]))
    return out
