"""host-side utilities mirroring pyro.util (messages, runtime parameters, timers)"""
