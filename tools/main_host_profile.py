"""cProfile of main.py's frame loop thread (synthetic, example.yaml grid, PLY outputs): where the HOST spends a frame."""
import cProfile, pstats, sys, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import main as M
argv = ['-c', 'configs/example.yaml', '-m', 'test', '--synthetic', '--frames', '24', '--no-npz', '--save-ply', '--io-spare-cus', '0', '--output-dir', '/tmp/avc_hp'] + sys.argv[1:]
M.main(['-c', 'configs/example.yaml', '-m', 'test', '--synthetic', '--frames', '2', '--no-npz', '--output-dir', '/tmp/avc_hp'])
pr = cProfile.Profile(); pr.enable()
M.main(argv)
pr.disable()
s = io.StringIO(); st = pstats.Stats(pr, stream=s); st.sort_stats('tottime').print_stats(28); print(s.getvalue()[:6000])
s = io.StringIO(); st.sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
