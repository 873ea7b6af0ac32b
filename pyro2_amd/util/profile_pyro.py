"""Nested wall-clock timers with the pyro/util/profile_pyro.py:15-135 surface
(TimerCollection.timer(name) -> Timer.begin()/end(), report())."""
import time


class Timer:
    def __init__(self, name, stack_count=0):
        self.name = name
        self.stack_count = stack_count
        self.is_running = False
        self.elapsed_time = 0.0
        self.start_time = 0.0

    def begin(self):
        self.start_time = time.time()
        self.is_running = True

    def end(self):
        self.elapsed_time += time.time() - self.start_time
        self.is_running = False


class TimerCollection:
    def __init__(self):
        self.timers = []

    def timer(self, name):
        """existing timer of that name, or a new one nested under the timers
        currently running"""
        for t in self.timers:
            if t.name == name:
                return t
        depth = sum(1 for t in self.timers if t.is_running)
        t = Timer(name, stack_count=depth)
        self.timers.append(t)
        return t

    def report(self):
        for t in self.timers:
            print(t.stack_count * "   " + t.name + ": ", t.elapsed_time)
