"""CPU oracle for the GigaPose hot path -- TEST INFRASTRUCTURE ONLY.

May be imported solely by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg (as the checker / reported baseline).  gigapose_amd/ never imports it.
"""
