#!/usr/bin/env python
'''Entry point with the reference's command line (reference main.py:551-750):
    python main.py -m train -c config.json -ne 5 -ds synth
See danet-tensorflow_amd/cli.py.'''
import __graft_entry__ as graft

if __name__ == '__main__':
    graft.load_package()
    from danet_amd import cli
    cli.main()
