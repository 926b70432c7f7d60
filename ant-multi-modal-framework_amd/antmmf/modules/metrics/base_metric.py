"""BaseMetric (reference: antmmf/modules/metrics/base_metric.py): name + calculate / collect / summarize protocol."""


class BaseMetric:
    def __init__(self, name, *args, **kwargs):
        self.name = name

    def calculate(self, sample_list, model_output, *args, **kwargs):
        raise NotImplementedError("'calculate' must be implemented in the child class")

    def collect(self, *args, **kwargs):
        raise NotImplementedError("'collect' must be implemented in the child class")

    def summarize(self, *args, **kwargs):
        raise NotImplementedError("'summarize' must be implemented in the child class")

    def __call__(self, *args, **kwargs):
        return self.calculate(*args, **kwargs)
